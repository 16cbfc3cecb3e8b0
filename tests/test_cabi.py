"""CPU-only checks of the drop-in boundary: libunite_hip.so loads, exports every symbol include/unite_hip.h declares,
the ctypes table mirrors the header, and the product path refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "unite_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(unite_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from unite_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with `make -C unite_amd/csrc` or __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 28
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/unite_hip.h but not exported by libunite_hip.so"


def test_ctypes_table_matches_header():
    from unite_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.unite_abi_version() == 2
    assert lib.unite_target_arch() == b"gfx950"
    # struct layout: the ctypes mirror lists the fields of the header's unite_gemm_args, in order, with matching kinds
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "unite_hip.h")).read(), flags=re.S)
    body = re.search(r"typedef struct unite_gemm_args \{(.*?)\} unite_gemm_args;", text, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        kind = "ptr" if "*" in decl else ("i64" if "int64_t" in decl else ("f32" if decl.startswith("float") else "i32"))
        for name in re.sub(r"^[a-z0-9_ ]*?\b(?=[A-Za-z_]+\s*(,|$))", "", decl.replace("*", " ")).split(","):
            fields.append((name.strip().split()[-1], kind))
    kinds = {ctypes.c_void_p: "ptr", ctypes.c_int32: "i32", ctypes.c_int64: "i64", ctypes.c_float: "f32"}
    assert [(n, kinds[t]) for n, t in _lib.GemmArgs._fields_] == fields
    assert ctypes.sizeof(_lib.GemmArgs) % 8 == 0 and len(fields) == 37


def test_comm_header_symbols_are_exported():
    """include/unite_comm.h (the RCCL side of SURVEY 8b: unite_comm_{init,allreduce_bucket,broadcast,destroy}) against libunite_comm.so and
    the ctypes table; loading it needs librccl, no GPU"""
    from unite_amd import _lib
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "unite_comm.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(unite_comm_[a-z0-9_]+)\s*\(", text)))
    assert names == sorted(_lib.COMM_SIGNATURES) and {"unite_comm_init", "unite_comm_allreduce_bucket", "unite_comm_broadcast",
                                                      "unite_comm_destroy"} <= set(names)
    assert os.path.exists(_lib.COMM_LIB_PATH), "build with `make -C unite_amd/csrc`"
    lib = _lib.load_comm()
    for n in names:
        assert hasattr(lib, n)
    assert lib.unite_comm_world() == 0                     # no communicator before unite_comm_init


def test_workspace_queries_run_without_gpu():
    from unite_amd import ops
    assert ops.layernorm_bwd_workspace(10240, 768) == (10240 // 16) * 3 * 768 * 4
    assert ops.colsum_workspace(1000, 776) > 0
    assert ops.grad_norm_workspace(88_005_888) > 0
    assert ops.gemm_colsum_workspace(10240, 3072) == 32768 + 80 * 3072 * 4          # counters header + one partial row per 128 output rows


def test_no_cpu_fallback():
    """device ops and models must fail loudly on CPU tensors instead of silently computing somewhere else"""
    from unite_amd import ops
    from unite_amd._lib import UniteHipError
    a = torch.zeros(16, 16, dtype=torch.bfloat16)
    with pytest.raises(UniteHipError):
        ops.gemm(a, a, torch.zeros(16, 16))
    with pytest.raises(UniteHipError):
        ops.layernorm_fwd(torch.zeros(4, 64), torch.ones(64), torch.zeros(64), 1e-6, torch.zeros(4, 64))
    from tests.test_model_gpu import build_tiny
    s, t = build_tiny()
    with pytest.raises(RuntimeError, match="MI355X"):
        s(torch.zeros(1, 3, 2, 32, 32), torch.zeros(1, 8, dtype=torch.bool))
    with pytest.raises(RuntimeError, match="MI355X"):
        t(torch.zeros(1, 3, 2, 32, 32))


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under unite_amd/ may import it"""
    for dp, _, fs in os.walk(os.path.join(ROOT, "unite_amd")):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)


def test_planner_choices_for_the_weight_gradients_of_the_step():
    """unite_gemm_plan (host arithmetic, no GPU): tile kernel and split factor of the stage-1 step's weight-gradient products (K = 10 240 visible
    tokens), alone on the GPU and beside the teacher (sharing 0.8 and 0.9).  The sharing-0.8 column is the set the overlapped step measured
    fastest with in rounds 2-3 (profiles/r03_ab_notes.txt: every forced alternative and the stand-alone-fitted cost model were slower); the
    0.9 column -- fewer slices still -- is what stage 1 runs with since the end of round 4 (engine_stage1.AheadStream.DEFAULT_SHARING has the
    same-call A/B) -- a planner edit that moves one of them should come with a new same-box A/B."""
    import ctypes as C
    from unite_amd import _lib
    lib = _lib.load()
    slabs = 16 * 3072 * 1024 * 4
    want = {  # (M, N): ((kind, S) alone, (kind, S) at sharing 0.8, at sharing 0.9)
        (2304, 768): ((1, 4), (2, 4), (2, 2)), (3072, 768): ((1, 3), (2, 3), (2, 2)), (768, 3072): ((1, 3), (2, 3), (2, 2)),
        (768, 768): ((1, 14), (1, 5), (1, 4)), (512, 768): ((1, 16), (1, 7), (1, 4))}
    if os.environ.get("UNITE_PLAN_MODEL", "2") != "2":
        pytest.skip("a non-default cost model is selected")
    for (M, N), plans in want.items():
        for w, exp in zip((0.0, 0.8, 0.9), plans):
            k, s = C.c_int32(), C.c_int32()
            assert lib.unite_gemm_plan(M, N, 10240, 1, 1, w, slabs, 0, C.byref(k), C.byref(s)) == 0
            assert (k.value, s.value) == exp, (M, N, w, k.value, s.value)
    k, s = C.c_int32(), C.c_int32()
    assert lib.unite_gemm_plan(768, 768, 10240, 1, 1, 0.8, 0, 0, C.byref(k), C.byref(s)) == 0 and s.value == 1       # no workspace: never split
    assert lib.unite_gemm_plan(768, 768, 10240, 1, 1, 1.5, slabs, 0, C.byref(k), C.byref(s)) < 0                      # weight outside [0, 1]


def test_grouped_gemm_refuses_the_fused_bias_sums():
    """unite_gemm_bf16_grouped runs the plain kernel form: a problem that asks for rowsum_a_out / colsum_out must be refused (UNITE_EINVAL),
    not answered with UNITE_OK and the sums unwritten (round-3 advisor finding).  Argument checking only: no device is touched."""
    import ctypes as C
    from unite_amd import _lib
    lib = _lib.load()

    def problem(**extra):
        g = _lib.GemmArgs()
        g.M, g.N, g.K, g.trans_a, g.trans_b = 256, 256, 512, 1, 1
        g.A, g.lda, g.B, g.ldb, g.out, g.ldc, g.out_f32 = 0x10000, 256, 0x20000, 256, 0x30000, 256, 1      # never dereferenced: refused first
        for k, v in extra.items():
            setattr(g, k, v)
        return g
    for bad in (dict(rowsum_a_out=0x40000), dict(colsum_out=0x40000)):
        arr = (_lib.GemmArgs * 2)(problem(), problem(**bad))
        assert lib.unite_gemm_bf16_grouped(arr, 2, None) == -1, bad
    # (a schedule hint that is not 0 / 1 cannot be expressed: plan_flags bit 3 IS the value)


def test_one_rccl_per_process():
    """libunite_comm.so binds RCCL at run time and must end up on the copy PyTorch ships (the one torch.distributed's "nccl" backend runs on):
    round 3's library LINKED /opt/rocm/lib/librccl.so beside the wheel's, two RCCL runtimes in one process (27.5 vs 20.4 ms per step in the
    one-rank rehearsal).  Checked without a GPU: the bound file, the dynamic section of the .so, and this process's memory map."""
    import subprocess
    import torch
    from unite_amd import _lib
    bound = os.path.realpath(_lib.comm_library())
    wheel = os.path.realpath(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    assert bound == wheel, (bound, wheel)
    needed = subprocess.run(["readelf", "-d", _lib.COMM_LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" not in needed, needed                      # nothing of RCCL is linked
    mapped = {os.path.realpath(l.split()[-1]) for l in open("/proc/self/maps") if "librccl" in l}
    assert mapped == {wheel}, mapped
