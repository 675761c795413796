"""ORACLE -- test infrastructure only (see sia_oracle.py / onepass.py / onepass_c.c headers).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; nothing under tissue_analysis_amd/ does.
"""
