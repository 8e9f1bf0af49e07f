# usage: scripts/bench_flag.sh <module> <Class> <attr>   -> bench.py with the class-level switch on/off/on/off (whole step incl. graph capture)
for rep in 1 2; do for flag in True False; do
python - "$@" $flag <<'PY' 2>/dev/null
import importlib, io, json, runpy, sys, contextlib
mod, cls, attr, flag = sys.argv[1:5]
setattr(getattr(importlib.import_module(mod), cls), attr, flag == "True")
sys.argv = ["bench.py", "--no-cpu-baseline"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("%s.%s=%s: %.2f ms/step  %.2f images/s" % (cls, attr, flag, d["ms_per_step"], d["value"]))
PY
done; done
