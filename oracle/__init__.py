"""ORACLE package: CPU restatements of the reference hot path. Test infrastructure only --
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from
the product package."""
