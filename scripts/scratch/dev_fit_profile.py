"""cProfile of a whole varGP fit (examples/one_cell_fit.py settings) -- where the host time goes."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["one_cell_fit.py"] + sys.argv[1:]
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.path.dirname(__file__), "..", "examples", "one_cell_fit.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:70]))
