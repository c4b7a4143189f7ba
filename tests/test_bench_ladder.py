"""bench.py --gpus N without a launcher: the parent walks a ladder of fresh child processes (driver / exchange rungs), each
with a wall-clock limit, and prints the line of the first rung that verifies -- carrying `fallback_from`.  CPU only: the rungs
are stand-in children (a stub runner, and real child processes for the timeout / kill path)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench   # noqa: E402


def test_ladder_order():
    lad = lambda *a: bench.ladder(bench.parse(["--gpus", "8"] + list(a)))
    assert lad() == [("torch", "rccl"), ("torch", "copy"), ("group", "copy")]
    assert lad("--exchange", "copy") == [("torch", "copy"), ("group", "copy")]
    assert lad("--driver", "group") == [("group", "copy")]          # the group's RCCL exchange is opt-in only
    assert lad("--driver", "group", "--exchange", "rccl") == [("group", "rccl"), ("group", "copy")]
    assert lad("--fallback", "0") == [("torch", "rccl")]


def test_rung_commands_are_fresh_children_of_the_right_kind():
    argv = ["--gpus", "4", "--steps", "20", "--warmup", "5", "--exchange=rccl", "--driver", "torch"]
    args = bench.parse(argv)
    t = bench.rung_command(args, "torch", "copy", argv)
    assert t[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in t and t[t.index("--nproc-per-node") + 1] == "4"
    assert "127.0.0.1" in t and t[-4:] == ["--driver", "torch", "--exchange", "copy"]
    assert t.count("--exchange") == 1 and "--exchange=rccl" not in t and t.count("--driver") == 1
    g = bench.rung_command(args, "group", "copy", argv)
    assert g[1].endswith("bench.py") and g[-4:] == ["--driver", "group", "--exchange", "copy"] and "--steps" in g


def test_first_rung_fails_second_is_printed_with_its_history(capsys):
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    args = bench.parse(argv)
    seen = []

    def runner(cmd, timeout_s, env):
        assert env.get("SVO_BENCH_CHILD") == "1"        # a rung never walks the ladder itself
        seen.append(cmd)
        ex = cmd[cmd.index("--exchange") + 1]
        drv = cmd[cmd.index("--driver") + 1]
        # the copy rung keeps its control plane off RCCL (what failed the rung before it may be RCCL itself)
        assert env.get("SVO_BENCH_BACKEND", "") == ("gloo" if (drv, ex) == ("torch", "copy") else "")
        if (drv, ex) == ("torch", "rccl"):
            return 1, None, "RuntimeError: NCCL error: unhandled system error"
        return 0, {"metric": "m", "value": 123.0, "n_gpus": 8, "verified": True, "driver": drv, "exchange": ex, "fallback_from": []}, ""

    rc = bench.launch_ranks(args, argv, runner=runner)
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
    assert rc == 0 and len(out) == 1 and len(seen) == 2
    line = json.loads(out[0])
    assert line["value"] == 123.0 and line["exchange"] == "copy" and line["driver"] == "torch"
    assert line["fallback_from"] == [{"driver": "torch", "exchange": "rccl", "failed": "exit code 1",
                                      "stderr_tail": "RuntimeError: NCCL error: unhandled system error"}]


def test_unverified_and_timed_out_rungs_fall_through_and_total_failure_is_a_line(capsys):
    argv = ["--gpus", "2"]
    args = bench.parse(argv)
    answers = iter([(0, {"value": 1.0, "verified": False}, "mismatch"), (None, None, ""), (3, None, "boom")])
    rc = bench.launch_ranks(args, argv, runner=lambda cmd, t, env: next(answers))
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
    line = json.loads(out[-1])
    assert rc == 1 and line["value"] is None and [f["failed"] for f in line["fallback_from"]] == [
        "verified: false", "timed out after %d s" % int(args.rung_timeout), "exit code 3"]


def test_run_rung_kills_a_hung_child_and_reads_the_last_json_line(tmp_path):
    hang = tmp_path / "hang.py"
    hang.write_text("import time, sys\nprint('starting', flush=True)\nsys.stderr.write('about to hang\\n'); sys.stderr.flush()\ntime.sleep(600)\n")
    t0 = time.time()
    rc, line, err = bench.run_rung([sys.executable, str(hang)], 1.5)
    assert rc is None and line is None and err == "about to hang" and time.time() - t0 < 30
    good = tmp_path / "good.py"
    good.write_text("import json\nprint('noise')\nprint(json.dumps({'value': 1}))\nprint(json.dumps({'value': 2, 'verified': True}))\n")
    rc, line, err = bench.run_rung([sys.executable, str(good)], 30)
    assert rc == 0 and line == {"value": 2, "verified": True}


def test_a_child_of_the_ladder_does_not_walk_it(monkeypatch):
    """main() only walks the ladder in the process the user started: a rung (SVO_BENCH_CHILD=1) or a rank of a launcher
    (RANK set) goes straight to the run."""
    called = []
    monkeypatch.setattr(bench, "launch_ranks", lambda *a, **k: called.append(1) or 0)
    monkeypatch.setenv("SVO_BENCH_CHILD", "1")
    try:
        bench.main(["--gpus", "2", "--driver", "group"])
    except BaseException:       # no GPU here: the run itself stops at "bench.py needs a GPU"
        pass
    assert not called


# ---- under a launcher (how the round driver starts N > 1) there is no parent to walk a ladder: the ranks try the exchange out in
# child processes BEFORE they touch the GPU and agree over a CPU group (bench.negotiate_exchange)
def _negotiate_worker(rank, world, port, fail_on, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench as b
    args = b.parse(["--gpus", str(world), "--steps", "20", "--warmup", "5"])
    seen = []

    def runner(cmd, timeout_s, env):
        ex = cmd[cmd.index("--exchange") + 1]
        seen.append((ex, env["MASTER_PORT"], env.get("SVO_BENCH_BACKEND", ""), env.get("SVO_BENCH_CHILD"), "--probe" in cmd and cmd[cmd.index("--probe") + 1]))
        assert not any(k.startswith("TORCHELASTIC_") for k in env)
        if (rank, ex) in fail_on:
            return (None, None, "hung in ncclCommInitRank") if ex == "rccl" else (1, None, "boom")
        return 0, ({"verified": True} if rank == 0 else None), ""

    exchange, backend, tried = b.negotiate_exchange(args, dist, torch, runner=runner)
    with open(os.path.join(out_dir, "r%d.json" % rank), "w") as f:
        json.dump({"exchange": exchange, "backend": backend, "tried": tried, "seen": seen, "initialized": dist.is_initialized()}, f)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def _negotiate(tmp_path, fail_on):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_negotiate_worker, args=(2, port, fail_on, str(tmp_path)), nprocs=2, join=True)
    return [json.load(open(tmp_path / ("r%d.json" % r))) for r in range(2)], port


def test_ranks_under_a_launcher_keep_the_rccl_gather_when_every_probe_is_clean(tmp_path):
    res, port = _negotiate(tmp_path, fail_on=set())
    for r in res:
        assert (r["exchange"], r["backend"], r["tried"]) == ("rccl", "nccl", []) and r["initialized"] is False
        assert [r["seen"][0][0]] + r["seen"][0][2:] == ["rccl", "", "1", "0"] and len(r["seen"]) == 1   # no probing inside the probe
    # a rendezvous of its own: a free port rank 0 picked, carried to every rank by the gloo group -- not the launcher's
    assert res[0]["seen"][0][1] == res[1]["seen"][0][1] != str(port)


def test_one_rank_s_probe_hangs_and_all_ranks_agree_on_the_copy_exchange(tmp_path):
    res, port = _negotiate(tmp_path, fail_on={(1, "rccl")})
    for rank, r in enumerate(res):
        assert (r["exchange"], r["backend"]) == ("copy", "gloo") and r["initialized"] is True    # the CPU group stays as control plane
        assert [t["exchange"] for t in r["tried"]] == ["rccl"] and r["tried"][0]["failed"].startswith("probe: ")
        assert [s[0] for s in r["seen"]] == ["rccl", "copy"] and r["seen"][1][2] == "gloo"
    assert res[0]["seen"][1][1] == res[1]["seen"][1][1] and res[0]["seen"][1][1] not in (str(port), res[0]["seen"][0][1])
    assert "timed out" in res[1]["tried"][0]["failed"] and "another rank" in res[0]["tried"][0]["failed"]


def test_nothing_probes_clean_the_run_goes_on_with_the_copy_exchange_and_says_so(tmp_path):
    res, _ = _negotiate(tmp_path, fail_on={(0, "rccl"), (1, "copy")})
    for r in res:
        assert (r["exchange"], r["backend"]) == ("copy", "gloo") and [t["exchange"] for t in r["tried"]] == ["rccl", "copy"]
