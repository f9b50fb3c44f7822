"""ORACLE package: CPU restatements of the reference's hot path, used ONLY as the checker.

Test infrastructure.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import anything from here; kp_gnn_amd (the product) never does and fails loudly without its HIP
library instead of falling back to this code.
"""
