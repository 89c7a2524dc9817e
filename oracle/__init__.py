"""oracle/ -- TEST INFRASTRUCTURE ONLY.  CPU (NumPy) restatement of the reference's
per-timestep grid update, used solely as the checker by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg.  The product (qingdai_amd) never imports it."""
