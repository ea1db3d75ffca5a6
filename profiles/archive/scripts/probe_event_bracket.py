"""Probe: what a hipEvent bracket adds to a short kernel (rpt_event_bracket in librp_mathtest.so)."""
import ctypes as C, os
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "commonroad-reactive-planner_amd", "lib", "librp_mathtest.so"))
lib.rpt_event_bracket.argtypes = [C.c_double, C.c_int, C.POINTER(C.c_double)]
out = (C.c_double * 2)()
for spin in (5.0, 15.0, 17.0, 50.0):
    assert lib.rpt_event_bracket(spin, 41, out) == 0
    print(f"empty bracket {out[0]:6.2f} us | bracket around a 465 x 256 kernel spinning {spin:5.1f} us: {out[1]:6.2f} us  (+{out[1] - spin:5.2f})")
