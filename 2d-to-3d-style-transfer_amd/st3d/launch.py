"""One process per GPU, started from a parent that never touches a GPU itself.

``python bench.py --gpus N`` (and the drop-in CLIs with ``--world_size N``) must stand on their own: the parent
starts N fresh children -- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT in their
environment, exactly what ``torch.distributed.run`` would set -- waits for them, relays rank 0's standard output and
fails if any child does.  No exec: a process that has initialised HIP must not be replaced, and the parent stays
GPU-free so nothing of it occupies a card.  If one rank dies the others are stopped (by the PIDs started here)
instead of waiting in a rendezvous that can never complete.

The reference has no counterpart (single process, single device: SURVEY.md 2.3).
"""
import os
import socket
import subprocess
import sys
import time


def under_launcher():
    """True inside a rank process (torchrun or spawn_ranks set WORLD_SIZE)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               GROUP_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # the host driver of this pool only supports dmabuf IPC: RCCL's peer mappings fail without it
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def spawn_ranks(world, argv, timeout=None, poll=0.05):
    """Run ``argv`` (a full command line) once per rank.  Rank 0's stdout is captured and returned; every other stream
    is inherited (so warnings of all ranks reach the caller's stderr).  Returns (returncode, rank0_stdout):
    returncode is 0 only if every rank exited 0, else the first non-zero code seen."""
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(argv, env=rank_env(r, world, port), stdin=subprocess.DEVNULL, text=True,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    t0, rc = time.time(), 0
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    try:
        alive = set(range(world))
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print("st3d.launch: rank %d exited with code %d; stopping the other ranks" % (r, code), file=sys.stderr)
            if rc != 0 or (timeout is not None and time.time() - t0 > timeout):
                if rc == 0:
                    rc = 124
                    print("st3d.launch: ranks still running after %.0f s; stopping them" % timeout, file=sys.stderr)
                break
            time.sleep(poll)
    finally:
        for p in procs:                      # only the exact PIDs started above
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=5)
    return rc, (out0[0] if out0 else "")


def self_launch(world, script, args):
    """Parent side of ``python <script> --gpus N``: spawn the ranks and return their exit code.  Rank 0's result line (its
    last stdout line that is a JSON object) goes to stdout alone, whatever else it printed (e.g. gloo's connection banner)
    to stderr -- the caller's contract is ONE line on stdout."""
    rc, out = spawn_ranks(world, [sys.executable, script] + list(args))
    lines = out.splitlines()
    result = next((k for k in range(len(lines) - 1, -1, -1) if lines[k].lstrip().startswith("{")), None)
    for k, line in enumerate(lines):
        print(line, file=sys.stdout if k == result else sys.stderr)
    sys.stdout.flush()
    if result is None and rc == 0:
        print("st3d.launch: rank 0 printed no result line", file=sys.stderr)
        rc = 1
    return rc
