import sys, runpy
name, val = sys.argv[1], int(sys.argv[2]); sys.argv = ["bench.py"] + sys.argv[3:]
sys.path.insert(0, ".")
import dmcnet_amd
dmcnet_amd._lib.check(dmcnet_amd._lib.load().dmc_set_option(name.encode(), val), "set")
runpy.run_path("bench.py", run_name="__main__")
