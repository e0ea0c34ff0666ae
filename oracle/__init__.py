"""CPU oracle for the hot path -- TEST INFRASTRUCTURE ONLY.

`oracle.c` (+ `ref.py`, its numpy front-end) is a plain-C restatement of the reference's
algorithm; `ref_torch.py` is the reference's CPU call sequence over torch CPU ops.  Both are
pinned to the reference by tests/test_oracle_golden.py.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package; cpu-vision_amd/ never does.
"""
