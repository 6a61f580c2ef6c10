"""TEST INFRASTRUCTURE -- CPU restatements of the reference's arithmetic (torch fp32), pinned by tests/golden/.
Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package."""
