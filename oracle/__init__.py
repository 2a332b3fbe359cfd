"""TEST INFRASTRUCTURE - CPU restatement of the reference's forward path (oracle/unimedvl_cpu.py, oracle/fp8.py) and the
scripts that pin it to the imported reference (oracle/gen_golden.py -> tests/golden/).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package; the product (unimedvl_amd/) never does."""
