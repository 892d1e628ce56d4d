#!/usr/bin/env python
"""A/B of one environment switch on the same box: runs `bench.py --no-cpu-baseline` alternately with VAR=a and
VAR=b (child processes) and prints images/sec of every run.   python tools/ab_env.py VAR a b [repeats] [bench args]"""
import json
import os
import subprocess
import sys


def main():
    var, a, b = sys.argv[1:4]
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    extra = sys.argv[5:]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for _ in range(reps):
        for v in (a, b):
            env = dict(os.environ, **{var: v})
            out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--no-cpu-baseline'] + extra, env=env,
                                 capture_output=True, text=True, timeout=600)
            line = [l for l in out.stdout.splitlines() if l.startswith('{')]
            if not line:
                print(var, v, 'FAILED', out.stderr[-400:], flush=True)
                continue
            d = json.loads(line[-1])
            print(f'{var}={v}: {d["value"]} img/s, {d["ms_per_step"]} ms', flush=True)


if __name__ == '__main__':
    main()
