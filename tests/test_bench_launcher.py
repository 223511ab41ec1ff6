"""bench.py's N > 1 entry as the driver uses it: `python bench.py --gpus N ...` with no torchrun around it.  The launcher must
start the ranks as a child torch.distributed.run (rendezvous at 127.0.0.1), relay rank 0's ONE JSON line and the exit code.
Rehearsed on CPU with --dry-run (gloo, no GPU work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, env=e, timeout=300)


def test_gpus2_self_launches_two_ranks():
    p = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                    # exactly ONE line on stdout: rank 0's JSON
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["dry_run"] is True


def test_single_rank_does_not_spawn():
    p = _run("--dry-run", "--config", "cfg1")
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads(p.stdout.strip())
    assert rec["n_gpus"] == 1 and rec["steps"] == 4      # cfg1's own step count


def test_child_failure_is_relayed_once_without_retry():
    # rank 1 of the CHILD exits non-zero (not an argparse error of the parent): the launcher relays a failure, prints no JSON line
    # and does NOT start the all-gather retry - that is reserved for a failed all-to-all probe
    p = _run("--gpus", "2", "--dry-run", env={"DRN_DRYRUN_FAIL": "1:3"})
    assert p.returncode != 0 and not p.stdout.strip()
    assert "one fresh run" not in p.stderr and "PROBE_FAILED" not in p.stderr


def test_probe_failure_retries_once_with_gather_and_says_so():
    p = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", env={"DRN_DRYRUN_FAIL": "0:probe"})
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stderr.count("one fresh run with DRN_SP_EXCHANGE=gather") == 1
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["exchange_fallback"] is True and rec["retried_exchange"] == "gather" and rec["first_attempt_rc"] not in (0, None)


def test_probe_failure_with_gather_already_selected_is_not_retried():
    # the hook only fires while the all-to-all exchange is selected, as the real probe: with gather nothing fails
    p = _run("--gpus", "2", "--dry-run", env={"DRN_DRYRUN_FAIL": "0:probe", "DRN_SP_EXCHANGE": "gather"})
    assert p.returncode == 0 and "one fresh run" not in p.stderr
    assert "exchange_fallback" not in json.loads(p.stdout.strip())


def test_hung_child_is_killed_at_the_limit():
    p = _run("--gpus", "2", "--dry-run", env={"DRN_DRYRUN_FAIL": "1:hang", "DRN_BENCH_TIMEOUT_S": "20"})
    assert p.returncode == 124 and not p.stdout.strip()


def test_busy_rendezvous_port_is_retried_on_a_fresh_one():
    # the port handed to torch.distributed.run is taken (here: by this test) before it can bind: no rank ever starts; the launcher
    # issues the same command again on another port and says so in the relayed line
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        sk.listen(1)
        p = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", env={"DRN_BENCH_FIRST_PORT": str(sk.getsockname()[1])})
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stderr.count("rendezvous port in use") == 1 and "one fresh run" not in p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["rendezvous_port_retries"] == 1 and "exchange_fallback" not in rec
