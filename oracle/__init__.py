"""ORACLE — TEST INFRASTRUCTURE ONLY (parity unpinned; see oracle/caffe_ref.py and DESIGN.md).

CPU restatement of the reference hot path used as the checker by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg.  The shipped package never imports it.
"""
