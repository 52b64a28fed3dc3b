"""oracle/ -- CPU restatement of the reference's FSR1/NIS hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (openvr_fsr_amd) never does.
"""
