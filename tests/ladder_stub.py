"""Stand-in for bench.py's GPU child in the launch-ladder tests (RGBDR_BENCH_CHILD_CMD): no torch, no GPU.
STUB_PLAN is a JSON list with one behaviour per rung:
  "ok"            rendezvous on MASTER_PORT, rank 0 prints a provisional and a final line, status 0
  "hang"          every rank sleeps for ever
  "fail"          every rank exits 5 (rank 0 prints an {"error": ...} line first)
  "provisional"   rank 0 prints the provisional line, then every rank sleeps for ever
  "rank1 dies"    rank 1 exits 7 at once, the others sleep for ever
"""
import json
import os
import socket
import sys
import time

rung = int(os.environ["RGBDR_BENCH_RUNG"])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["RGBDR_BENCH_ROLE"] == "rank" and not [k for k in os.environ if k.startswith("TORCHELASTIC_")]
plan = json.loads(os.environ["STUB_PLAN"])[rung]
log = os.environ.get("STUB_LOG")
if log:
    with open(os.path.join(log, "rung%d.rank%d" % (rung, rank)), "w") as f:
        json.dump({"argv": sys.argv[1:], "port": os.environ["MASTER_PORT"]}, f)


def rendezvous():
    """rank 0 listens on the rung's port, every other rank connects: proves that all ranks were given the same, free port"""
    port = int(os.environ["MASTER_PORT"])
    if rank == 0:
        with socket.socket() as s:
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            s.bind(("127.0.0.1", port))
            s.listen(world)
            s.settimeout(20)
            for _ in range(world - 1):
                c, _ = s.accept()
                c.close()
    else:
        for _ in range(200):
            try:
                socket.create_connection(("127.0.0.1", port), timeout=1).close()
                return
            except OSError:
                time.sleep(0.05)
        sys.exit(9)


line = {"metric": "stub", "value": 100.0 + rung, "n_gpus": world, "config": {"argv": sys.argv[1:]}}
if plan == "hang":
    time.sleep(1e6)
if plan == "fail":
    if rank == 0:
        print(json.dumps({"error": "stub failure on rung %d" % rung}), flush=True)
    sys.exit(5)
if plan == "rank1 dies":
    if rank == 1:
        sys.exit(7)
    time.sleep(1e6)
rendezvous()
if rank == 0:
    print("chatter that is not JSON", flush=True)
    print(json.dumps(dict(line, provisional=True)), flush=True)
if plan == "provisional":
    time.sleep(1e6)
if rank == 0:
    print(json.dumps(dict(line, legs="done")), flush=True)
sys.exit(0)
