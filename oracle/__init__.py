"""CPU oracle for the chnsPyramid + acfDetect path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package; nothing under acf_amd/ does.
"""
