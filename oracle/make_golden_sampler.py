#!/usr/bin/env python
"""Golden index lists for unite_amd.data.DistributedSampler, produced by the reference's own sampler
(src/datasets/distributed.py:81-163, imported by file path; it needs torch only).  Run in the build container:

    python oracle/make_golden_sampler.py      ->  tests/golden/sampler.json

Test infrastructure: the reference does not travel to the GPU box, the fixture does."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("UNITE_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "sampler.json")


def main():
    spec = importlib.util.spec_from_file_location("ref_distributed", os.path.join(REF, "src/datasets/distributed.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    cases = []
    for n, replicas, shuffle, seed, drop_last, reps, epochs in [
            (10, 1, True, 0, False, 1, (0, 1)),
            (10, 4, True, 0, False, 1, (0, 3)),
            (10, 4, True, 7, True, 1, (0,)),
            (10, 4, False, 0, False, 3, (0,)),
            (37, 8, True, 0, False, 2, (0, 5)),
            (37, 8, True, 0, True, 2, (2,)),
            (3, 8, True, 1, False, 1, (0,)),         # fewer samples than ranks: the padding wraps around more than once
            (3, 8, False, 0, False, 2, (0,)),
            (64, 2, True, 123, False, 1, (0, 1, 2))]:
        ds = list(range(n))
        for epoch in epochs:
            per_rank = []
            for rank in range(replicas):
                s = m.DistributedSampler(ds, num_replicas=replicas, rank=rank, shuffle=shuffle, seed=seed, drop_last=drop_last,
                                         repetitions=reps)
                s.set_epoch(epoch)
                idx = list(iter(s))
                assert len(idx) == len(s)
                per_rank.append(idx)
            cases.append(dict(n=n, num_replicas=replicas, shuffle=shuffle, seed=seed, drop_last=drop_last, repetitions=reps, epoch=epoch,
                              indices=per_rank))
    json.dump(dict(source="src/datasets/distributed.py:81-163 (reference DistributedSampler with `repetitions`)", cases=cases),
              open(OUT, "w"), separators=(",", ":"))
    print(OUT, len(cases), "cases")


if __name__ == "__main__":
    sys.exit(main())
