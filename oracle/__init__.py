"""CPU oracle for the UniVS hot path -- TEST INFRASTRUCTURE, never imported by `univs_amd/`.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
  ops_ref.c      plain-C restatement of the operator semantics (MSDA forward, mask decode, attention
                 mask rule, Swin window attention); built by oracle/Makefile into oracle/_build/
  c_ops.py       ctypes/numpy binding of ops_ref.c
  torch_ref.py   fp32 CPU restatement of the module-level path (Swin, pixel decoder, UniVS decoder)
  ref_harness.py / gen_golden.py   dev-container-only: import the real reference and emit tests/golden/
"""
