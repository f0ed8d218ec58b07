"""cProfile of the reference-API loop's host side (the profiler's own overhead inflates everything; the proportions are what counts)"""
import cProfile, pstats, os, sys, runpy, io
sys.argv = [sys.argv[0], sys.argv[1] if len(sys.argv) > 1 else "256", "6000"]
here = os.path.dirname(os.path.abspath(__file__))
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(here, "api_step_histogram.py"), run_name="__main__")
finally:
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
s2 = io.StringIO()
st = pstats.Stats(pr, stream=s2)
st.print_callers("_cuda_getDeviceCount")
st.print_callers("is_available")
print(s2.getvalue()[:3000])
