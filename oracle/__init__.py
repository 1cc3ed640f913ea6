"""CPU oracle for the VideoNavQA video-question fusion path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
