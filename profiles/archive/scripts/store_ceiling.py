"""Practical HBM *write* ceiling of this MI355X for the state-block stream: plain coalesced streaming stores."""
import ctypes as C, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(REPO, "commonroad-reactive-planner_amd", "lib", "librp_mathtest.so"))
lib.rpt_store_ceiling.restype = C.c_double
lib.rpt_store_ceiling.argtypes = [C.c_size_t, C.c_int, C.c_int]
for nbytes, label in ((25920960, "cfg2 state blocks (25.9 MB)"), (11436096256, "cfg5 state blocks (11.4 GB)")):
    for wide in (1, 0):
        ms = lib.rpt_store_ceiling(nbytes, wide, 10)
        print(f"{label:32s} {'16 B/lane' if wide else ' 8 B/lane'}: {ms*1e3:10.1f} us  {nbytes/ms/1e6:8.1f} GB/s  ({nbytes/ms/1e6/8000*100:.1f} % of 8 TB/s)")
