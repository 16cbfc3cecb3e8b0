"""state_dict shape tables used to build weights in tests without importing the reference,
and the tiny configurations the golden fixtures were generated with (oracle/make_golden.py)."""
from oracle import umt_oracle as O

TINY_T = O.TeacherCfg(input_resolution=32, patch_size=16, width=128, layers=3, heads=2, output_dim=64, clip_return_layers=(1, 2))
TINY_S = O.StudentCfg(img_size=32, patch_size=16, embed_dim=128, depth=3, num_heads=2, num_frames=2, tubelet_size=1,
                      clip_decoder_embed_dim=128, clip_output_dim=64, clip_return_layers=(1, 2))
TINY_V = O.VitCfg(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, num_classes=5, all_frames=4)


def student_shapes(cfg):
    D, H = cfg.embed_dim, int(cfg.embed_dim * cfg.mlp_ratio)
    p = cfg.patch_size
    s = [("encoder.patch_embed.proj.weight", (D, 3, cfg.tubelet_size, p, p)), ("encoder.patch_embed.proj.bias", (D,))]
    for i in range(cfg.depth):
        b = f"encoder.blocks.{i}."
        s += [(b + "norm1.weight", (D,)), (b + "norm1.bias", (D,)),
              (b + "attn.q_bias", (D,)), (b + "attn.v_bias", (D,)),
              (b + "attn.qkv.weight", (3 * D, D)), (b + "attn.proj.weight", (D, D)), (b + "attn.proj.bias", (D,)),
              (b + "norm2.weight", (D,)), (b + "norm2.bias", (D,)),
              (b + "mlp.fc1.weight", (H, D)), (b + "mlp.fc1.bias", (H,)),
              (b + "mlp.fc2.weight", (D, H)), (b + "mlp.fc2.bias", (D,))]
    s += [("encoder.norm.weight", (D,)), ("encoder.norm.bias", (D,))]
    for k in range(len(cfg.clip_return_layers)):
        d = f"clip_decoder.{k}."
        s += [(d + "head.weight", (cfg.clip_output_dim, cfg.clip_decoder_embed_dim)), (d + "head.bias", (cfg.clip_output_dim,)),
              (d + "norm.weight", (cfg.clip_output_dim,)), (d + "norm.bias", (cfg.clip_output_dim,))]
    return s


def teacher_shapes(cfg):
    W = cfg.width
    g = cfg.input_resolution // cfg.patch_size
    s = [("class_embedding", (W,)), ("positional_embedding", (g * g + 1, W)), ("proj", (W, cfg.output_dim)),
         ("conv1.weight", (W, 3, cfg.kernel_size, cfg.patch_size, cfg.patch_size)),
         ("ln_pre.weight", (W,)), ("ln_pre.bias", (W,))]
    for i in range(cfg.layers):
        b = f"transformer.resblocks.{i}."
        s += [(b + "attn.in_proj_weight", (3 * W, W)), (b + "attn.in_proj_bias", (3 * W,)),
              (b + "attn.out_proj.weight", (W, W)), (b + "attn.out_proj.bias", (W,)),
              (b + "ln_1.weight", (W,)), (b + "ln_1.bias", (W,)),
              (b + "mlp.c_fc.weight", (4 * W, W)), (b + "mlp.c_fc.bias", (4 * W,)),
              (b + "mlp.c_proj.weight", (W, 4 * W)), (b + "mlp.c_proj.bias", (W,)),
              (b + "ln_2.weight", (W,)), (b + "ln_2.bias", (W,))]
    s += [("ln_post.weight", (W,)), ("ln_post.bias", (W,))]
    return s


def vit_shapes(cfg):
    D, H, p = cfg.embed_dim, 4 * cfg.embed_dim, cfg.patch_size
    s = [("patch_embed.proj.weight", (D, 3, cfg.tubelet_size, p, p)), ("patch_embed.proj.bias", (D,))]
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        s += [(b + "norm1.weight", (D,)), (b + "norm1.bias", (D,)), (b + "attn.q_bias", (D,)), (b + "attn.v_bias", (D,)),
              (b + "attn.qkv.weight", (3 * D, D)), (b + "attn.proj.weight", (D, D)), (b + "attn.proj.bias", (D,)),
              (b + "norm2.weight", (D,)), (b + "norm2.bias", (D,)), (b + "mlp.fc1.weight", (H, D)), (b + "mlp.fc1.bias", (H,)),
              (b + "mlp.fc2.weight", (D, H)), (b + "mlp.fc2.bias", (D,))]
    s += [("fc_norm.weight", (D,)), ("fc_norm.bias", (D,)), ("head.weight", (cfg.num_classes, D)), ("head.bias", (cfg.num_classes,))]
    return s
