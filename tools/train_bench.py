"""Time one training step of the bench workload and print the per-kernel-class breakdown (development tool)."""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, "bench.py", "--steps", "8", "--warmup", "3", "--no-cpu-baseline"] + sys.argv[1:],
                     capture_output=True, text=True).stdout.strip().splitlines()
if not out:
    raise SystemExit("bench.py produced no output")
d = json.loads(out[-1])
print(d["value"], d["ms_per_step"], d["kernel_class_ms_per_step"])
