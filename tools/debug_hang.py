import faulthandler, runpy, sys
faulthandler.dump_traceback_later(int(sys.argv[1]), exit=True)
script = sys.argv[2]
sys.argv = sys.argv[2:]
runpy.run_path(script, run_name="__main__")
