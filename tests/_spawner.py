"""A helper process that starts OTHER processes on behalf of the GPU test session. It is started by tests/conftest.py at
session start, before the pytest process has made any HIP call, and never touches the GPU itself: the rank processes of
the multi-rank GPU tests are its children, not children of a process that has initialised the GPU (on this pool such a
process must not fork/exec). Protocol: one JSON object per line on stdin
    {"argv": [...], "envs": [{...}, ...], "timeout": seconds, "cwd": "..."}
starts len(envs) copies of argv (each with os.environ + its env), waits for all of them, and answers with one line
    {"rc": [...], "tail": ["last 4000 characters of stdout+stderr", ...]}
on stdout. EOF on stdin ends it."""
import json
import os
import subprocess
import sys
import tempfile
import time


def run(job):
    procs, files = [], []
    for env in job["envs"]:
        f = tempfile.TemporaryFile(mode="w+")
        files.append(f)
        procs.append(subprocess.Popen(job["argv"], env={**os.environ, **{k: str(v) for k, v in env.items()}}, cwd=job.get("cwd"),
                                      stdout=f, stderr=subprocess.STDOUT, start_new_session=True))
    deadline = time.time() + float(job.get("timeout", 600))
    rcs = [None] * len(procs)
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if time.time() > deadline or any(rc not in (None, 0) for rc in rcs):  # one rank failed: the others would wait for it forever
            time.sleep(2.0)
            for i, p in enumerate(procs):
                if p.poll() is None:
                    p.kill()  # exactly the processes started above
                    p.wait()
                    rcs[i] = -9
                else:
                    rcs[i] = p.returncode
            break
        time.sleep(0.2)
    tails = []
    for f in files:
        f.seek(0)
        tails.append(f.read()[-4000:])
        f.close()
    return {"rc": rcs, "tail": tails}


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        try:
            out = run(json.loads(line))
        except Exception as e:  # noqa: BLE001
            out = {"rc": [-1], "tail": [repr(e)]}
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
