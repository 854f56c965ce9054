"""oracle/ — TEST INFRASTRUCTURE ONLY.

CPU restatement of the MoleculeSDE pretrain hot path (restate.py) plus the harness that
runs the reference's own files verbatim (ref_loader.py + standins/) to generate golden
vectors.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; nothing under moleculesde_amd/ does.
"""
