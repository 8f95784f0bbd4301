import sys, runpy
name, val = sys.argv[1], int(sys.argv[2]); sys.argv = ["bench.py"] + sys.argv[3:]
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import dmcnet_amd
dmcnet_amd._lib.check(dmcnet_amd._lib.load().dmc_set_option(name.encode(), val), "set")
runpy.run_path(__import__("os").path.join(sys.path[0], "bench.py"), run_name="__main__")
