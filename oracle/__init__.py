"""CPU restatements of the reference's hot-path algorithms -- TEST INFRASTRUCTURE ONLY.

Nothing under oracle/ may be imported by the product package (speech2text_amd);
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as
the checker / timed CPU baseline.  Each function cites the reference file:line
(relative to the reference tree) it restates.  Pinning status per module is in
the module header and in DESIGN.md.
"""
