"""Runs test functions one by one and asks hipGetLastError() after each (and after a gc pass): which one leaves an error behind?
usage: python scripts/probes/stale_hip_error_after_tests.py tests/test_halo_gpu.py"""
import ctypes, gc, importlib.util, inspect, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = ctypes.c_char_p


def ask(where):
    e = hip.hipGetLastError()
    print(f"{'STALE ' + str(e) + ' ' + hip.hipGetErrorString(e).decode() if e else 'clean'}: {where}", flush=True)


spec = importlib.util.spec_from_file_location("t", sys.argv[1])
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
for name, fn in inspect.getmembers(mod, inspect.isfunction):
    if not name.startswith("test_"):
        continue
    marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
    argsets = [()]
    if marks:
        argsets = [a if isinstance(a, tuple) else (a,) for a in marks[0].args[1]]
    for a in argsets:
        try:
            fn(*a)
        except Exception as ex:   # noqa
            print("  raised", type(ex).__name__, str(ex)[:100])
        ask(f"{name}{a} returned")
        gc.collect(); ask(f"{name}{a} + gc")
