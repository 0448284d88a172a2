#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

  python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 and no WORLD_SIZE in the environment: this process starts
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same args>
as a child BEFORE it touches the GPU, relays rank 0's JSON line and exits with the child's code; fewer than N visible
GPUs -> non-zero exit and a message.  Launched by torch.distributed.run from outside (the driver), WORLD_SIZE must equal --gpus.

A "step" = one pass of the hot path (model forward -> decode -> per-class NMS [-> all-gather of the decoded
boxes when N > 1]) over one batch of synthetic frames that is already resident in HBM.  By default TWO batches are in flight
(--in-flight 2, yolo_fastest_amd.BatchPipeline): consecutive steps are issued round-robin on two HIP streams, each with its own
engine, so a batch's late per-frame stages run beside the next batch's early machine-filling ones.  Which two streams is decided by
an untimed measurement before the warm-up (BatchPipeline.tune_streams: the overlap of two streams depends on the hardware queues the
runtime happened to give them -- 253-261 k frames/s on an unlucky pair, 291-295 k on the others).
All K steps (and exchanges) complete inside the timed region, `ms_per_step` is elapsed / K.  The one-step-at-a-time figure (two half-batch
lanes) is reported beside it (`one_batch_in_flight`), as are the per-pass model / post-process times (`forward_chain.forward_ms`, `post_ms`).
Workload at N = 1: BASELINE.json configs[1], "YOLO-Fastest 320x256 batch=256 fp32, synthetic frames"
(u8 ~ Uniform{0..255} i.i.d., x = (u8-128)/255, seed = rank).  N > 1: the same per GPU (weak
scaling, frames are independent units; one RCCL all-gather of the fixed-capacity box records).
Rank 0 prints ONE JSON line.

The `roofline` object is the PHYSICAL roof of the dominant launch (README "Reading the bench line"):
  * the launch's flops, split by the pipe they run on in the plan (matrix cores / vector ALU, yf_op_info_ex), over its
    average duration (HIP events on the launch stream), against the dense peak of that pipe and type;
  * its HBM bytes from the PMC counters (profiles/pmc_traffic.json, taken with tools/refresh_profiles.sh at the source hash
    recorded in the file; null when the file is from other sources) over the same duration against 8 TB/s;
  * `bound` names the larger of the two fractions ("hbm" when the counter bytes per second exceed half of the 6.3 TB/s a
    streaming kernel reaches).
SURVEY.md 8(d)'s layer-granular figure (what an UNFUSED network would move) is kept under `layer_granular_equiv`; a fused
plan exceeds 1.0 on it by construction, so it is not a roofline fraction.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd")

BYTES_PER_FRAME = {256: 42_575_360, 512: 170_301_440}   # layer-granular algorithmic bytes, fp32 (fp16: half)
FLOPS_PER_FRAME = {256: 236_442_880, 512: 945_771_520}
# MI355X_MICROARCH.md, chip-level parameters
HBM_PEAK_GBS = 8000.0          # HBM3E spec
HBM_STREAM_GBS = 6290.0        # measured float4 copy
FP32_PEAK_TF = 157.3           # vector fp32 == fp32-input MFMA (64 FLOP/clk/SIMD each; they share the issue rate)
F16_MFMA_PEAK_TF = 2500.0      # dense fp16/bf16 MFMA


def source_hash():
    """sha256 over the sources of the inference kernels and the engine: identifies the build a profile under profiles/ belongs to.
    (The training-step kernels, yf_train_* / yf_loss_*, are not launched by this benchmark and are left out, so that work on them does
    not orphan the PMC traffic figures.)"""
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")) and not f.startswith(("yf_train_", "yf_loss_")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def cpu_baseline(seconds=12.0):
    """The oracle (CPU restatement of the reference's detect.py path; kind "port") timed on this host:
    batch 1 over the 20 bundled frames, model and post-process like detect.py:151-171, repeated for ~`seconds`."""
    import numpy as np
    import torch
    from oracle import backbone_oracle as bo
    from oracle import post_oracle as po
    import yolo_fastest_amd as yf
    io = yf.io_params_for(256)
    sd = bo.load_state_dict(os.path.join(PKG, "assets", "weights", "yolo_fastest_256x320_epoch28.pth"))
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_256.npz"))
    xs = [bo.preprocess(g["input_u8"][i]) for i in range(20)]
    default_threads = torch.get_num_threads()
    runs = []
    for threads in sorted({1, 8, default_threads}):  # batch-1 convs of this size do not scale with cores
        torch.set_num_threads(threads)
        bo.forward(sd, xs[0])  # warm-up
        n, t_model, t_post = 0, 0.0, 0.0
        t_end = time.time() + seconds / 3
        while time.time() < t_end:
            for x in xs:
                t0 = time.time()
                hl, hs = bo.forward(sd, x)
                t1 = time.time()
                po.post_process((hl.numpy()[0], hs.numpy()[0]), io["anchors"], io["input_shape"][:2])
                t2 = time.time()
                t_model += t1 - t0; t_post += t2 - t1; n += 1
        runs.append((n / (t_model + t_post), threads, n, 1e3 * t_model / n, 1e3 * t_post / n))
    torch.set_num_threads(default_threads)
    best = max(runs)
    return {"value": round(best[0], 2), "unit": "frames/s", "cores": best[1], "kind": "port",
            "sample": "the 20 bundled test_data frames at 320x256, batch 1 (detect.py:151-171: model, then decode+sort+NMS), "
                      f"repeated ~{seconds / 3:.0f} s per thread setting; best of "
                      + "; ".join(f"{t} threads: {f:.1f} frames/s ({n} frames, model {m:.2f} ms + post {p:.2f} ms)"
                                  for f, t, n, m, p in runs)
                      + f"; host has {os.cpu_count()} logical CPUs"}


def _run_child(cmd, timeout):
    """subprocess.run for the profiler passes, with the WHOLE process group ended on a timeout (the profiler's child is the python
    process that holds the GPU).  -> (returncode, combined output tail)"""
    import signal
    p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         start_new_session=True)
    try:
        out, _ = p.communicate(timeout=timeout)
        return p.returncode, out[-300:]
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        p.communicate()
        return -9, "timed out after %d s" % timeout


ISSUE_COUNTERS = ("SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS")
SIMDS, PEAK_GHZ = 1024, 2.4


def live_pmc(res, batch, dtype, frames, kmax, with_issue=False):
    """Counters of every launch of a single-lane forward pass of that workload, measured IN THIS RUN: child runs of this script under
    `rocprofv3 --pmc` (counters with --kernel-trace only; FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md's HBM section
    prescribes; with_issue: a third pass with the issue-side SQ counters), started before this process touches the GPU.  Units and
    corrections of that guide: FETCH_SIZE / WRITE_SIZE are KiB of 64-byte fabric requests; on gfx950 FETCH_SIZE reports half of the bytes of
    wide (16 B / lane) coalesced streaming reads, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Launches
    are matched to the plan's ops by order (one lane, one batch in flight: deterministic).
    Returns ({op name: HBM bytes per launch}, {op name: {counter: value per launch}}, note): ({}, {}, why) when the profiler is not usable."""
    import csv
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {}, {}, "rocprofv3 not found"
    work = tempfile.mkdtemp(prefix="yf_pmc_", dir="/tmp")
    vals = {}
    try:
        passes = [("FETCH_SIZE",), ("WRITE_SIZE",)] + ([ISSUE_COUNTERS] if with_issue else [])
        ops = None
        for pi, counters in enumerate(passes):
            d = os.path.join(work, "p%d" % pi)
            cmd = [prof, "--pmc"] + list(counters) + ["--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-variants", "--no-configs",
                   "--no-live-traffic", "--no-train", "--no-extras", "--in-flight", "1", "--lanes", "1", "--res", str(res), "--batch", str(batch),
                   "--dtype", dtype, "--frames", frames, "--kmax", str(kmax), "--launch-repeats", "1", "--dump-ops", os.path.join(work, "ops.json")]
            rc, tail = _run_child(cmd, 120)
            path = os.path.join(d, "p_counter_collection.csv")
            if rc != 0 or not os.path.exists(path):
                return {}, {}, "rocprofv3 --pmc %s child failed (rc %d): %s" % (counters[0], rc, tail.replace("\n", " | "))
            with open(os.path.join(work, "ops.json")) as f:
                ops = json.load(f)
            # kernel dispatches per launch: 1, except a residual chain issued block by block at small batches (yf_op_dispatches)
            disp = [int(o.get("dispatches", 1)) for o in ops]
            first = [sum(disp[:i]) for i in range(len(ops))]
            n = sum(disp)
            by = {}
            with open(path) as f:   # (kernels instantiated on _Float16 come out mangled: the demangler does not know DF16_)
                for x in csv.DictReader(f):
                    k = x["Kernel_Name"]
                    if ("yf::" in k or k.startswith("_ZN2yf")) and "post_kernel" not in k and "nms_sorted" not in k and "spin" not in k:
                        by.setdefault(int(x["Dispatch_Id"]), {})[x["Counter_Name"]] = float(x["Counter_Value"])
            ids = sorted(by)
            if n == 0 or len(ids) < n or len(ids) % n:
                return {}, {}, "PMC rows (%d) do not divide into forward passes of %d launches" % (len(ids), n)
            np_ = len(ids) // n
            for c in counters:
                vals[c] = [sum(by[ids[q * n + first[i] + d]].get(c, 0.0) for q in range(np_) for d in range(disp[i])) / np_ for i in range(len(ops))]
        traffic, issue = {}, {}
        for i, o in enumerate(ops):
            traffic[o["name"]] = traffic.get(o["name"], 0.0) + vals["FETCH_SIZE"][i] * 1024 * 2 + vals["WRITE_SIZE"][i] * 1024
            if with_issue:
                issue[o["name"]] = {c: vals[c][i] for c in ISSUE_COUNTERS}
        return {k: int(round(v)) for k, v in traffic.items()}, issue, None
    except Exception as e:   # the profiler is evidence, not product: never fail the benchmark over it
        return {}, {}, "live PMC pass failed: %r" % (e,)
    finally:
        shutil.rmtree(work, ignore_errors=True)


TRAIN_FLOPS_PER_EXAMPLE = {256: 2 * (3 * 118_221_440 - 1_474_560)}   # forward + backward-data (conv0 needs none) + weight-gradient MACs x 2


def live_train_traffic(batch, iters=3):
    """HBM bytes of ONE training iteration at this batch size, by the counters, in this run: tools/train_bench.py under `rocprofv3 --pmc`
    (FETCH_SIZE and WRITE_SIZE passes, the units and the gfx950 doubling of FETCH_SIZE as in live_traffic) summed over every kernel of
    the process -- ours and torch's -- and divided by the iterations.  Also returns the kernel that holds the largest share of the
    kernel time of that (counter-slowed) run.  -> (bytes per iteration or None, dominant kernel or None, its share, note)"""
    import csv
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, None, None, "rocprofv3 not found"
    work = tempfile.mkdtemp(prefix="yf_pmct_", dir="/tmp")
    total, times = 0.0, {}
    try:
        for counter, scale in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
            d = os.path.join(work, counter)
            cmd = [prof, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable,
                   os.path.join(ROOT, "tools", "train_bench.py"), "--batch", str(batch), "--steps", str(iters), "--warmup", "0"]
            rc, tail = _run_child(cmd, 240)
            path = os.path.join(d, "p_counter_collection.csv")
            if rc != 0 or not os.path.exists(path):
                return None, None, None, "rocprofv3 --pmc %s child failed (rc %d): %s" % (counter, rc, tail.replace("\n", " | "))
            with open(path) as f:
                for x in csv.DictReader(f):
                    if x["Counter_Name"] != counter:
                        continue
                    total += float(x["Counter_Value"]) * scale
                    if counter == "FETCH_SIZE":
                        k = x["Kernel_Name"].split("(")[0]
                        times[k] = times.get(k, 0.0) + (int(x["End_Timestamp"]) - int(x["Start_Timestamp"]))
        dom = max(times, key=times.get) if times else None
        return total / iters, dom, (times[dom] / sum(times.values()) if dom else None), None
    except Exception as e:
        return None, None, None, "live PMC pass failed: %r" % (e,)
    finally:
        shutil.rmtree(work, ignore_errors=True)


TRAIN_WARMUP = 8


def train_records(dev, pmc):
    """The training iteration (train.py:111-132: zero_grad, train-mode forward, two-head loss, backward, Adam) on synthetic frames at the
    reference's batch size (16, _config.py:41) and at 256: examples/s, and the physical roofs of the iteration -- flops over the fp32
    issue peak, counter-measured HBM bytes over 8 TB/s."""
    import numpy as np
    import torch
    import yolo_fastest_amd as yf
    from yolo_fastest_amd import training, validation as val
    io = yf.io_params_for(256)
    H, W = io["input_shape"][:2]
    out = []
    for batch, steps in ((16, 30), (256, 6)):
        torch.manual_seed(0)
        m = yf.YoloFastest(io)
        m.initialize_weights()
        m = m.to(dev).train()
        x = (torch.rand(batch, 1, H, W) - 0.5).to(dev)
        rng = np.random.default_rng(0)
        t = np.zeros((batch, 64, 6), np.float32)
        for b in range(batch):
            k = 1 + b % 6
            t[b, :k, 0:2] = rng.uniform(0.05, 0.95, (k, 2)); t[b, :k, 2:4] = rng.uniform(0.03, 0.4, (k, 2))
            t[b, :k, 4] = rng.integers(0, 3, k); t[b, :k, 5] = 255.0
        td = torch.from_numpy(t).to(dev)
        crit = [val.YOLOLossV3(io["anchors"][i], 3, io["input_shape"], dev, model=m) for i in range(2)]
        opt = training.Adam(m.parameters(), lr=0.001)
        for _ in range(TRAIN_WARMUP):                          # (the trainer captures its passes as graphs once their pointers repeat: iterations 2-5)
            training.train_step(m, crit, opt, x, td)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = training.train_step(m, crit, opt, x, td)[0]
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        flops = TRAIN_FLOPS_PER_EXAMPLE[256] * batch
        rec = {"workload": "train.py:111-132 iteration (zero_grad, train-mode forward, two-head loss, backward, Adam), 320x256, synthetic frames and targets",
               "batch": batch, "steps": steps, "warmup": TRAIN_WARMUP, "dtype": "f32", "value": round(batch / dt, 1), "unit": "examples/s",
               "ms_per_iteration": round(1e3 * dt, 3), "loss_finite": bool(torch.isfinite(loss.detach()).item()),
               "roofline": {"flops_per_iteration": flops, "compute": {"achieved": round(flops / dt / 1e12, 2), "peak": FP32_PEAK_TF, "unit": "TFLOP/s",
                                                                     "frac": round(flops / dt / 1e12 / FP32_PEAK_TF, 4)},
                            "hbm": None, "bound": None, "dominant_kernel": None}}
        tb, dom, share, note = pmc.get(batch, (None, None, None, "not measured at this batch size (kernel-latency-bound: ~475 launches per iteration)"))
        if tb is not None:
            gbs = tb / dt / 1e9
            rec["roofline"]["hbm"] = {"traffic": int(tb), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                      "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/train_bench.py, all kernels"}
            rec["roofline"]["bound"] = "hbm" if gbs / HBM_PEAK_GBS > rec["roofline"]["compute"]["frac"] else "fp32 issue"
            rec["roofline"]["dominant_kernel"] = {"name": dom[:100], "share_of_kernel_time_in_the_counter_pass": round(share, 3)}
        elif note:
            rec["roofline"]["note"] = note
        out.append(rec)
        del m, opt, crit, x, td
        torch.cuda.empty_cache()
    return out


def self_launch(n_gpus, share_gpu=False):
    """--gpus N > 1 without a launcher: start the N ranks as a CHILD process tree (never exec: this process must not have
    touched the GPU, and it has not -- device_count() does not initialise it) and relay its output."""
    import torch
    have = torch.cuda.device_count()
    if have < n_gpus and not (share_gpu and have >= 1):
        sys.stderr.write(f"bench.py: --gpus {n_gpus} but only {have} GPU(s) are visible on this host; refusing to run a smaller "
                         f"job under that label\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def plan_rank_cpus(local_world, allowed, sysfs="/sys", visible=None):
    """The CPU slice of EVERY local rank, decided by ONE scheme for all of them: if sysfs resolves the NUMA node of every
    rank's GPU (KFD topology node -> PCI address -> numa_node -> cpulist; no GPU call) and every node has at least two allowed CPUs per
    rank hanging off it, each rank gets a share of its GPU's node; if ANY rank cannot be resolved, every rank gets a contiguous equal slice
    of the allowed set.  (Mixing the two per rank could hand a fallback rank CPUs inside another rank's NUMA share.)  Assumes KFD topology
    order == HIP device order, narrowed by the first of HIP/ROCR/CUDA_VISIBLE_DEVICES when `visible` is given.  Returns (slices, how)."""
    allowed = sorted(allowed)

    def gpu_numa_nodes():
        out, base = [], f"{sysfs}/class/kfd/kfd/topology/nodes"
        for n in sorted(os.listdir(base), key=int):
            props = dict(l.split()[:2] for l in open(f"{base}/{n}/properties") if len(l.split()) >= 2)
            if int(props.get("simd_count", "0")) == 0:
                continue        # a CPU node
            loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
            bdf = "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7)
            with open(f"{sysfs}/bus/pci/devices/{bdf}/numa_node") as f:
                out.append(int(f.read()))
        return out

    def cpulist(node):
        cpus = []
        for part in open(f"{sysfs}/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus += list(range(int(a), int(b or a) + 1))
        return cpus

    slices, how = None, "contiguous equal slices of the allowed CPUs (all ranks)"
    try:
        nodes = gpu_numa_nodes()
        if visible:
            nodes = [nodes[int(v)] for v in visible.split(",") if v.strip().isdigit() and int(v) < len(nodes)]
        if len(nodes) >= local_world and all(n >= 0 for n in nodes[:local_world]):
            numa = []
            for r in range(local_world):
                peers = [q for q in range(local_world) if nodes[q] == nodes[r]]
                cpus = [c for c in cpulist(nodes[r]) if c in set(allowed)]
                per = len(cpus) // len(peers)
                if per < 2:
                    numa = None
                    break
                k = peers.index(r)
                numa.append(cpus[k * per:(k + 1) * per])
            if numa is not None:
                slices, how = numa, "share of the NUMA node of each rank's GPU (sysfs; all ranks)"
    except (OSError, ValueError, IndexError, KeyError):
        slices = None
    if slices is None:
        per = len(allowed) // local_world
        slices = [allowed[r * per:(r + 1) * per] for r in range(local_world)]
    flat = [c for sl in slices for c in sl]
    assert len(flat) == len(set(flat)) and all(sl for sl in slices), "rank CPU slices must be disjoint and non-empty"
    return slices, how


def pin_rank_to_cpus(local_rank, local_world, sysfs="/sys"):
    """One rank = one GPU = one slice of the host's CPUs: at 8 ranks every host thread issues ~26 k launches/s and
    must neither migrate nor share a core with another rank's.  Called BEFORE the first GPU call, so the HIP runtime's and RCCL's helper
    threads inherit the mask.  Every process computes ALL ranks' slices (plan_rank_cpus: one scheme for all, disjoint by assertion) and
    takes its own.  Returns a small dict for the JSON line (or None: nothing pinned)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None
    if local_world < 2 or len(allowed) < 2 * local_world:
        return None
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    try:      # pinning is an optimisation: whatever goes wrong on a topology this was never run on, the benchmark runs unpinned
        slices, how = plan_rank_cpus(local_world, allowed, sysfs, vis)
        mine = slices[local_rank]
        os.sched_setaffinity(0, mine)
    except Exception:
        return None
    return {"cpus": len(mine), "first": mine[0], "last": mine[-1], "how": how}


def launch_roofline(o, dtype, traffic, issue=None):
    """Physical roof of one launch.  fp32: MFMA and vector FMAs share ONE issue path on this part -- measured in isolation in round 5
    (tools/coissue_probe.hip, profiles/r05_mfma_valu_coissue.txt: a wave streaming v_mfma_f32_16x16x4_f32 / 4x4x1 / 32x32x2 and a wave
    streaming v_fma_f32 / v_pk_fma_f32 / v_mov_b32 on the same SIMD reach 1.00-1.02 of ONE issue rate together, in either age order; the
    fp16 MFMA control reaches 1.84-2.00) -- so the floor is (all flops) / 157.3 TF, the SUM of the two, not the larger.  fp16: the matrix cores run beside the vector ALU, the floor is the larger of the two.
    issue (optional): the launch's issue-side counters from live_pmc().  Then `bound` is COUNTER-BACKED instead of the larger of two flop
    ratios: the share of the launch's time the VALU needs to issue its instructions (SQ_INSTS_VALU x 4 cycles per SIMD -- address math,
    operand splitting and epilogues included, not only FMAs), the matrix pipe's busy share (SQ_VALU_MFMA_BUSY_CYCLES), the HBM share
    (counter bytes at 8 TB/s) and the share its waves are parked at s_waitcnt / barriers (SQ_WAIT_ANY / SQ_WAVE_CYCLES): the largest
    pipe names the bound if it holds at least half of the time or more than the parked share, else the launch is latency-bound."""
    t = o["ms"] * 1e-3
    dtype = o.get("kernel_dtype", dtype)   # a f16x3 engine runs the launches without a split-operand kernel in exact fp32
    issued_frac = None
    if dtype != "f32":
        # split operands (f16x3): three fp16 MFMAs per k-group are ISSUED for one fp32-class product.  `achieved` / `frac` count the USEFUL
        # flops (one product per k-group) over the peak of the pipe they run on; the issued work decides which pipe is the busier one and is
        # reported beside it as `issued_frac` (round 5 printed the issued figure as `frac`: 3x too high for every f16x3 launch)
        mult = 3 if dtype == "f16x3" else 1
        t_mfma_issued = o["mfma_flops"] * mult / (F16_MFMA_PEAK_TF * 1e12)
        t_valu = o["valu_flops"] / (FP32_PEAK_TF * 1e12)
        on_mfma = t_mfma_issued >= t_valu
        peak_tf = F16_MFMA_PEAK_TF if on_mfma else FP32_PEAK_TF
        flops = o["mfma_flops"] if on_mfma else o["valu_flops"]
        floor = flops / (peak_tf * 1e12)
        issued_frac = max(t_mfma_issued, t_valu) / t
    else:
        floor = o["flops"] / (FP32_PEAK_TF * 1e12)
        on_mfma = o["mfma_flops"] >= o["valu_flops"]
        peak_tf, flops = FP32_PEAK_TF, o["flops"]
    r = {"compute_frac": floor / t, "on_mfma": on_mfma, "peak_tf": peak_tf, "achieved_tf": flops / t / 1e12, "hbm_frac": None,
         "hbm_gbs": None}
    if issued_frac is not None:
        r["issued_frac"] = issued_frac
    if traffic is not None:
        r["hbm_gbs"] = traffic / t / 1e9
        r["hbm_frac"] = r["hbm_gbs"] / HBM_PEAK_GBS
    hbm_bound = r["hbm_gbs"] is not None and r["hbm_gbs"] > 0.5 * HBM_STREAM_GBS
    r["bound"] = "hbm" if hbm_bound else ("mfma" if on_mfma else "valu")
    r["frac"] = r["hbm_frac"] if hbm_bound else r["compute_frac"]
    if issue:
        sec = SIMDS * PEAK_GHZ * 1e9
        c = {"valu": issue["SQ_INSTS_VALU"] * 4 / sec / t, "mfma": issue["SQ_VALU_MFMA_BUSY_CYCLES"] / sec / t,
             "hbm": r["hbm_frac"] or 0.0, "lds": issue["SQ_ACTIVE_INST_LDS"] / max(issue["SQ_WAVE_CYCLES"], 1.0)}
        parked = issue["SQ_WAIT_ANY"] / max(issue["SQ_WAVE_CYCLES"], 1.0)
        r["counters"] = dict({k + "_frac": round(v, 3) for k, v in c.items()}, parked_frac=round(parked, 3))
        pipes = dict(c)
        if dtype == "f32":     # fp32 MFMAs and VALU instructions share the issue slots: ONE pipe, named after where the launch's flops are
            pipes = {("mfma" if on_mfma else "valu"): c["valu"] + c["mfma"], "hbm": c["hbm"], "lds": c["lds"]}
            r["counters"]["fp32_issue_frac"] = round(c["valu"] + c["mfma"], 3)
        pipe = max(pipes, key=pipes.get)
        # `bound` is counter-backed; achieved / peak / frac stay the ROOFLINE figures above (useful flops over the peak of the pipe they
        # run on, or counter bytes over the HBM peak): a launch whose VALU is busy issuing address math, operand splits and epilogues
        # around its MFMAs is "valu"-bound at a low flop fraction -- that gap is the finding, not a rounding of it
        r["bound"] = pipe if (pipes[pipe] >= 0.5 or pipes[pipe] >= parked) else "latency"
        r["frac"] = r["hbm_frac"] if r["bound"] == "hbm" else r["compute_frac"]
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle", type=int, default=0, help="extra untimed steps of the headline loop in front of the W warm-up steps (off by default: "
                    "the contract's W is the only untimed work; the line reports warmup_effective = W + settle)")
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--res", type=int, default=256, choices=[256, 512])
    ap.add_argument("--chunk", type=int, default=0, help="frames per pass of the layer chain (0 = whole batch)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="batches in flight: consecutive steps are issued round-robin on this many streams, each with its own engine "
                         "(yolo_fastest_amd.BatchPipeline); 1 = one step at a time; 0 (default) = 2")
    ap.add_argument("--lanes", type=int, default=0, help="concurrent streams over the chunks of ONE batch (1..4); 0 = 1 with several "
                                                        "batches in flight, else 2")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16", "f16x3"],
                    help="f16 = fp16 storage + single-operand fp16 MFMA; f16x3 = fp32 storage + split-operand fp16 MFMA (fp32-class accuracy: "
                         "BASELINE configs[2] within its stated 2e-2)")
    ap.add_argument("--frames", default="noise", choices=["noise", "fixtures"],
                    help="noise: u8 ~ U{0..255} (BASELINE configs[1]); fixtures: the reference's 20 bundled frames tiled to the "
                         "batch (SURVEY.md 8(d) config 2 'realistic': dark IR frames, >= 1 box per frame)")
    ap.add_argument("--dense", action="store_true",
                    help="SURVEY.md 8(d) config 5: synthetic dense head logits (~1200 candidates, ~260 survivors per 640x512 frame) are "
                         "added to the heads of the noise frames, to stress decode + sort + NMS; use with --res 512 --batch 64 --kmax 1024")
    ap.add_argument("--branches", type=int, default=-1, choices=[-1, 0, 1],
                    help="1: the small head's launches on a side stream of their lane; -1 (default) = 0")
    ap.add_argument("--kmax", type=int, default=64)
    ap.add_argument("--exchange-at-1", action="store_true",
                    help="rehearsal of the N > 1 code path on one GPU: a 1-rank RCCL group and the all-gather of every step's records")
    ap.add_argument("--share-gpu", action="store_true",
                    help="REHEARSAL of the N > 1 code path where only one GPU exists: the N ranks all use GPU 0 and exchange their records over gloo "
                         "(launcher, rank / world checks, per-rank seeds, CPU pinning, barriers, max-over-ranks timing, gather, rank 0's line) -- "
                         "never a performance figure: the line says so (`rehearsal`)")
    ap.add_argument("--pin-cpus", type=int, nargs=2, default=None, metavar=("RANK", "OF"),
                    help="rehearse the multi-GPU CPU pinning on one GPU: pin this process like local rank RANK of OF ranks")
    ap.add_argument("--regions", type=int, default=5,
                    help="timed regions of --steps steps each, run back to back in the same loop after ONE warm-up: `value` is the first "
                         "(the contract's K steps), `repeat_values` lists them all (the box-to-box and run-to-run spread in the record)")
    ap.add_argument("--headline-only", action="store_true",
                    help="stop after the headline's timed regions and print a short line (kernel traces of exactly the headline loop: tools/refresh_profiles.sh)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra f16x3 measurement of the same workload")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure the HBM bytes per launch in this run (two rocprofv3 --pmc child passes before the benchmark); "
                         "roofline.traffic then comes from profiles/pmc_traffic.json if it was taken at this build, else null")
    ap.add_argument("--no-train", action="store_true", help="skip the training-iteration records (the `training` array)")
    ap.add_argument("--no-configs", action="store_true", help="skip the other single-GPU BASELINE.json configurations (the `configs` array)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra records of the default run: `fixtures` (the 20 bundled frames tiled to the batch), `u8` (the same "
                         "batch entering as u8 through the fused pre-process), `batch1` (latency of one frame), `exchange_rehearsal`")
    ap.add_argument("--launch-repeats", type=int, default=4,
                    help="per-launch timing (forward_chain.per_launch, roofline.launch_ms): back-to-back launches of each kernel between the two "
                         "HIP events that time it; 1 = one launch per event pair (the counter passes, which match dispatches to ops by order, use 1)")
    ap.add_argument("--dump-ops", default=None, help="write the launch names of one forward pass to this JSON file")
    ap.add_argument("--dump-records", default=None,
                    help="rank 0 writes the LAST step's detection records of all frames (N > 1: as gathered over RCCL) to this .npz")
    args = ap.parse_args()
    args.launch_repeats = max(1, min(64, args.launch_repeats))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(self_launch(args.gpus, args.share_gpu))
    world = int(env_world) if env_world is not None else 1
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; "
                         f"use --nproc-per-node == --gpus (or run plain `python bench.py --gpus N`, which launches the ranks itself)")

    # rank -> CPU slice, before the first GPU call of this process (N = 1: the whole host stays available)
    affinity = None
    if world > 1:
        affinity = pin_rank_to_cpus(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    elif args.pin_cpus:
        affinity = pin_rank_to_cpus(args.pin_cpus[0], args.pin_cpus[1])

    # HBM traffic of every launch, by the counters, in THIS run (children; this process has not touched the GPU yet)
    live, live_note = ({}, "not requested")
    # (not when this process itself runs under a profiler: its preloaded tool library has initialised the GPU already, and a process
    # that has must not start other programs on this pool)
    profiled = any(k.startswith(("ROCPROFILER_", "ROCP_", "ROCPROF")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if profiled:
        args.no_live_traffic = True
        live_note = "this process runs under a profiler: no nested counter passes"
    live_issue, cfg2_pmc = {}, {}
    default_wl = args.res == 256 and args.dtype == "f32" and not args.dense and args.frames == "noise"
    if world == 1 and not args.no_live_traffic and not args.dense and not args.exchange_at_1:
        live, live_issue, live_note = live_pmc(args.res, args.batch, args.dtype, args.frames, args.kmax, with_issue=True)
        if not args.no_configs and default_wl and live_note is None:   # (a profiler that failed once is not asked again)
            for dt in ("f16x3", "f16"):                # BASELINE configs[2]: HBM bytes and issue counters of the 640x512 fp16-pipe passes
                cfg2_pmc[dt] = live_pmc(512, 128, dt, "noise", 64, with_issue=True)
                if cfg2_pmc[dt][2] is not None:
                    break

    train_pmc = {}
    if world == 1 and not args.no_train and not args.no_live_traffic and args.res == 256 and args.dtype == "f32" and not args.dense and live_note is None:
        train_pmc[256] = live_train_traffic(256)

    import numpy as np
    import torch
    import torch.distributed as dist
    import yolo_fastest_amd as yf
    from yolo_fastest_amd import dist as yfd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.device_count() < (world if world > 1 and not args.share_gpu else 1):
        raise SystemExit(f"bench.py: {world} rank(s) but only {torch.cuda.device_count()} GPU(s) visible")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP extension is the only implementation (no CPU fallback)")
    dev = torch.device("cuda", 0 if args.share_gpu else local_rank)
    torch.cuda.set_device(dev)
    multi = world > 1 or args.exchange_at_1
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1 and "MASTER_ADDR" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
        if args.share_gpu:
            dist.init_process_group("gloo")       # (RCCL refuses two ranks on one GPU; gloo carries GPU tensors through the host)
        else:
            dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: RCCL world size {dist.get_world_size()} != --gpus {args.gpus}")

    WNAME = {256: "yolo_fastest_256x320_epoch28.pth", 512: "yolo_fastest_512x640_epoch27.pth"}

    def workload(res, batch, frames, dense):
        """Synthetic input of one configuration, resident in HBM: (io_params, x [batch,1,H,W], dense head field or None)."""
        io_ = yf.io_params_for(res)
        H_, W_ = io_["input_shape"][:2]
        g = torch.Generator(device="cpu").manual_seed(rank)
        if frames == "fixtures":
            fx = np.load(os.path.join(ROOT, "tests", "golden", "golden_%d.npz" % res))["input_u8"]
            u8 = torch.from_numpy(fx[(np.arange(batch) + rank * batch) % len(fx)])   # rank r continues where rank r - 1 stopped
        else:
            u8 = torch.randint(0, 256, (batch, H_, W_), generator=g, dtype=torch.uint8)
        x_ = ((u8.float() - 128.0) / 255.0)[:, None].contiguous().to(dev)
        syn_ = None
        if dense:  # per cell/anchor: t_conf ~ N(-1, 1.5^2) (~25 % pass), t_xy ~ N(0,1), t_wh ~ N(0,0.5^2), classes ~ N(0,2^2); seed = frame
            parts = ([], [])
            for f in range(batch):
                gg = np.random.default_rng(rank * batch + f)
                for i, (h, w) in enumerate(((H_ // 16, W_ // 16), (H_ // 32, W_ // 32))):
                    t = np.empty((3, 8, h, w), np.float32)
                    t[:, 0:2] = gg.normal(0.0, 1.0, (3, 2, h, w)); t[:, 2:4] = gg.normal(0.0, 0.5, (3, 2, h, w))
                    t[:, 4] = gg.normal(-1.0, 1.5, (3, h, w)); t[:, 5:8] = gg.normal(0.0, 2.0, (3, 3, h, w))
                    parts[i].append(t.reshape(24, h, w))
            syn_ = tuple(torch.from_numpy(np.stack(p)).to(dev) for p in parts)
        return io_, x_, syn_

    io, x, syn = workload(args.res, args.batch, args.frames, args.dense)
    wname = WNAME[args.res]
    H, W = io["input_shape"][:2]
    n_total = args.batch * world
    # batches in flight, automatic: two, on streams chosen by measurement (BatchPipeline.tune_streams): 320x256, 20 steps: 290-294 k
    # frames/s against 281-283 k one at a time (on an unlucky stream pair 253-261 k); 640x512 batch 128: 72.6 k against 69.7 k fp32,
    # 91.4 k against 85.4 k f16x3 (round 2's "no gain at 640x512" was such an unlucky pair)
    in_flight = args.in_flight if args.in_flight > 0 else 2
    lanes = args.lanes if args.lanes else (1 if in_flight > 1 else 2)
    # the small head in line: since it is ONE launch (fusion level 2) the side stream no longer pays at this batch size
    # (tools/fusion_ab.py, one batch at a time on two lanes: 280 k frames/s in line, 273 k on the side stream)
    branches = args.branches if args.branches >= 0 else 0
    cur = {"x": x, "syn": syn, "kmax": args.kmax, "n_total": n_total}   # the workload timed() / forward() run on

    def make(dtype, lanes_, branches_, io_=None, res_=None):
        io_ = io_ or io
        m = yf.YoloFastest(io_).to(dev).eval()
        m.chunk, m.lanes, m.branches = args.chunk, lanes_, branches_
        m.precision = dtype
        m.load_state_dict(torch.load(os.path.join(PKG, "assets", "weights", WNAME[res_ or args.res]), map_location=dev))
        p = yf.YOLO_post_process(io_["conf_thre"], io_["nms_thre"], io_["num_anchors"], io_["num_cls"], io_["anchors"], io_["input_shape"]).bind(m)
        return m, p

    def forward(m):
        with torch.no_grad():
            if cur["x"].dtype == torch.uint8 and cur["x"].dim() == 4 and m.input_channel == 1:
                pred = m.forward_bgr_u8(cur["x"], io["input_shape"])
            else:
                pred = m.forward_u8(cur["x"], io["input_shape"]) if cur["x"].dtype == torch.uint8 else m(cur["x"])
        if cur["syn"] is not None:   # the net's own logits on noise are ~no detections: the synthetic field replaces them
            pred = (pred[0] * 0 + cur["syn"][0], pred[1] * 0 + cur["syn"][1])
        return pred

    REC = ("counts", "boxes", "scores", "cls", "src")

    def timed(m, p, depth, steps, warmup, exchange, regions=1, region_times=None, settle=0):
        """W untimed + K timed steps; a step = model -> decode -> NMS over the resident batch [-> all-gather].  depth > 1: consecutive
        steps are issued round-robin on `depth` streams, each with its own engine (yolo_fastest_amd.BatchPipeline): a batch's late,
        per-frame stages run beside the next batch's early, machine-filling ones.  Every step -- and every exchange -- completes
        inside the timed region (device synchronisation + barrier on both sides).  Returns (seconds, last step's records)."""
        # exchange: the post-process writes ONE packed record block per batch (yf_decode_nms_packed) and the all-gather sends that buffer
        pipe = yf.BatchPipeline(m, p, depth=depth, kmax=cur["kmax"], lanes=m.lanes, branches=m.branches, packed=bool(exchange)) if depth > 1 else None
        if pipe is not None:
            pipe.tune_streams(cur["x"])      # untimed: which streams the batches overlap best on (pipeline.BatchPipeline.tune_streams)
        gather = (lambda out: yfd.all_gather_detections_async(out, cur["n_total"])) if exchange else None

        def run(n):
            last, pend = None, []
            for _ in range(n):
                if pipe is not None:
                    # (dense configurations: the synthetic logit field replaces the heads between model and post-process, inside the
                    #  batch's own stream: pipeline.BatchPipeline.submit `mid`)
                    splice = (lambda pred: (pred[0] * 0 + cur["syn"][0], pred[1] * 0 + cur["syn"][1])) if cur["syn"] is not None else None
                    last = pipe.submit(cur["x"], then=gather, mid=splice)
                    if gather is not None:
                        pend.append(last.extra)
                else:
                    raw = p.detect_raw(forward(m), kmax=cur["kmax"], packed=bool(exchange))
                    last = raw
                    if gather is not None:
                        pend.append(gather(raw))
                while len(pend) > 2 * depth:     # the exchange of step k runs on RCCL's stream while later steps compute
                    pend.pop(0).wait()
            gathered = None
            for h in pend:
                gathered = h.wait()
            if gathered is not None:
                cur["gathered"] = gathered
            if pipe is not None:
                pipe.drain()
                return last.synchronize()
            return last

        grouped = multi or bool(exchange)
        if settle:
            run(settle)      # untimed and outside W: the card's clocks and the host's caches settle (a cold first region read 6 % low once)
        run(warmup)
        first = None
        for reg in range(max(1, regions)):     # region 0 is THE timed region (exactly K steps); the others repeat it for the spread
            torch.cuda.synchronize(dev)
            if grouped:
                dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            raw_r = run(steps)
            torch.cuda.synchronize(dev)
            if grouped:
                dist.barrier()
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            if grouped:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            if region_times is not None:
                region_times.append(float(tt.item()))
            if first is None:
                first = (float(tt.item()), raw_r)
        return first

    def same_detections(a, b):
        if not torch.equal(a["counts"], b["counts"]):
            return False
        valid = torch.arange(a["cls"].shape[1], device=dev)[None, :] < a["counts"][:, None]
        return all(torch.equal(a[k][valid], b[k][valid]) for k in ("boxes", "cls", "src"))

    def post_mean_ms(m, p, kmax, reps=10):
        """decode + sort + NMS of one batch: mean of `reps` warmed repetitions on the same logits (HIP events on the launch stream)."""
        pred = forward(m)
        for _ in range(3):
            p.detect_raw(pred, kmax=kmax)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(reps):
            p.detect_raw(pred, kmax=kmax)
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / reps

    model, post = make(args.dtype, lanes, branches)
    region_s = []
    elapsed, raw = timed(model, post, in_flight, args.steps, args.warmup, multi, regions=args.regions, region_times=region_s, settle=args.settle)
    if args.headline_only:
        if rank == 0:
            print(json.dumps({"metric": "frames/sec end-to-end (headline loop only)", "value": round(n_total * args.steps / elapsed, 1), "unit": "frames/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
                              "repeat_values": [round(n_total * args.steps / t, 1) for t in region_s], "dtype": args.dtype, "in_flight": in_flight,
                              "source_hash": source_hash()}))
        if multi:
            dist.destroy_process_group()
        return

    # model / post-process split and the one-batch-at-a-time figure (two half-batch lanes, the small head on its side stream): measured
    # after the headline's timed region, same process
    m1, p1 = (model, post) if in_flight == 1 else make(args.dtype, 2, 0)
    es = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    fwd_ms = post_ms = 0.0
    for _ in range(10):
        p1.detect_raw(forward(m1), kmax=args.kmax)
    torch.cuda.synchronize(dev)
    for _ in range(10):
        es[0].record(); pred = forward(m1); es[1].record(); raw1 = p1.detect_raw(pred, kmax=args.kmax); es[2].record()
        torch.cuda.synchronize(dev)
        fwd_ms += es[0].elapsed_time(es[1]) / 10; post_ms += es[1].elapsed_time(es[2]) / 10
    single = other = None    # the other scheduling mode of the same workload, same loop, same run
    if in_flight > 1 and world == 1:
        e1, r1 = timed(m1, p1, 1, args.steps, args.warmup, False)
        single = {"in_flight": 1, "lanes": 2, "branches": 0, "value": round(args.batch * args.steps / e1, 1), "unit": "frames/s",
                  "ms_per_step": round(1e3 * e1 / args.steps, 4), "detections_identical": bool(same_detections(raw, r1))}
    elif world == 1 and syn is None and args.in_flight == 0:
        m2f, p2f = make(args.dtype, 1, 0)
        e1, r1 = timed(m2f, p2f, 2, args.steps, args.warmup, False)
        other = {"in_flight": 2, "lanes": 1, "branches": 0, "value": round(args.batch * args.steps / e1, 1), "unit": "frames/s",
                 "ms_per_step": round(1e3 * e1 / args.steps, 4), "detections_identical": bool(same_detections(raw, r1))}
        del m2f, p2f

    # The same workload on the split-operand fp16-MFMA variant (fp32 storage, fp32-class accuracy), measured in the SAME
    # run with the same loop: an extra object of the JSON line, never the headline `value` (which is the dtype named by --dtype).
    variant = None
    if world == 1 and args.dtype == "f32" and not args.no_variants and not args.dense:
        m2, p2 = make("f16x3", lanes, branches)
        e2, raw2 = timed(m2, p2, in_flight, args.steps, args.warmup, False)
        with torch.no_grad():
            a, b = model(x), m2(x)
        variant = {"dtype": "f16x3", "what": "fp32 storage, pointwise/dense GEMMs on the fp16 matrix pipe with split (hi + lo) operands, fp32 accumulate",
                   "in_flight": in_flight, "value": round(args.batch * args.steps / e2, 1), "unit": "frames/s",
                   "ms_per_step": round(1e3 * e2 / args.steps, 4),
                   "max_abs_logit_diff_vs_f32_on_this_batch": round(max(float((a[0] - b[0]).abs().max()), float((a[1] - b[1]).abs().max())), 6),
                   "detections_identical_to_f32_on_this_batch": bool(same_detections(raw, raw2))}
        del m2, p2

    # What already exists, in front of the driver (same process, after the headline, never `value`):
    extras = None
    if world == 1 and not args.no_extras and default_wl and args.batch == 256 and not args.exchange_at_1:
        extras = {}
        saved = dict(cur)

        def stem_us(m, xin):
            return round(1e3 * m.profile(xin, reps=5, launch_repeats=args.launch_repeats)[0]["ms"], 2)

        # (a) SURVEY.md 8(d).2 "realistic": configs[1] on the reference's 20 bundled frames tiled to 256 -- every frame has >= 1 box, so
        #     decode + sort + NMS do real work (the noise frames of the headline give ~0.02 survivors per frame)
        _, xf, _ = workload(256, args.batch, "fixtures", False)
        cur.update(x=xf, syn=None)
        ef, rawf = timed(model, post, in_flight, args.steps, args.warmup, False)
        cf = rawf["counts"].cpu().numpy()
        extras["fixtures"] = {"workload": "configs[1] on the reference's 20 bundled test_data frames tiled to the batch (SURVEY.md 8(d).2 'realistic')",
                              "value": round(args.batch * args.steps / ef, 1), "unit": "frames/s", "ms_per_step": round(1e3 * ef / args.steps, 4),
                              "steps": args.steps, "warmup": args.warmup, "in_flight": in_flight,
                              "survivors_per_frame_mean": round(float(np.clip(cf, 0, None).mean()), 3),
                              "frames_with_a_detection": int((cf > 0).sum()), "post_ms": round(post_mean_ms(m1, p1, args.kmax), 4)}
        # (b) SURVEY.md 8(f).1: the same noise batch entering as u8 -- Detect_YOLO.__pre_process's arithmetic (detect.py:107-129) fused
        #     into the stem's loads (yf_forward_u8) -> decode -> NMS; 1 byte per pixel read instead of 4
        g8 = torch.Generator(device="cpu").manual_seed(rank)
        u8 = torch.randint(0, 256, (args.batch, H, W), generator=g8, dtype=torch.uint8).to(dev)      # the headline's frames before (u8 - 128) / 255
        cur.update(x=u8, syn=None)
        eu, rawu = timed(model, post, in_flight, args.steps, args.warmup, False)
        with torch.no_grad():
            hu = model.forward_u8(u8, io["input_shape"])
            hx = model(saved["x"])
        t_u8 = []
        for _ in range(5):   # the stem launch alone, u8 input: HIP events around one pass's first launch are not exposed for this entry,
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # so time whole single-lane passes
            torch.cuda.synchronize(dev)
            e0.record(); m1.forward_u8(u8, io["input_shape"]); e1.record()
            torch.cuda.synchronize(dev)
            t_u8.append(e0.elapsed_time(e1))
        t_f32 = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            e0.record(); m1(saved["x"]); e1.record()
            torch.cuda.synchronize(dev)
            t_f32.append(e0.elapsed_time(e1))
        extras["u8"] = {"workload": "configs[1]'s frames entering as uint8 [256,256,320]: pre-process fused into the first kernel's loads "
                                    "(yf_forward_u8, detect.py:107-129) -> decode -> NMS",
                        "value": round(args.batch * args.steps / eu, 1), "unit": "frames/s", "ms_per_step": round(1e3 * eu / args.steps, 4),
                        "steps": args.steps, "warmup": args.warmup, "in_flight": in_flight,
                        "heads_bit_identical_to_the_f32_input_path": bool(torch.equal(hu[0], hx[0]) and torch.equal(hu[1], hx[1])),
                        "detections_identical_to_the_headline": bool(same_detections(raw, rawu)),
                        "forward_ms_two_lanes": {"u8_input": round(sorted(t_u8)[2], 4), "f32_input": round(sorted(t_f32)[2], 4)},
                        "stem_launch_us": {"u8_input": stem_us(model, u8), "f32_input": stem_us(model, saved["x"])}}
        # (b2) SURVEY.md 8 row A0 in full: the frames as cv2.imread returns them -- uint8 [256,480,640,3], BGR, a size that is neither the
        #      net's nor twice it -- through cvtColor(BGR2GRAY) + cv2.resize's INTER_LINEAR on the device (yf_forward_bgr_u8: one extra
        #      pass, HBM-bound byte work) -> the fused (v - 128) / 255 entry -> decode -> NMS
        gb = torch.Generator(device="cpu").manual_seed(rank + 1)
        bgr = torch.randint(0, 256, (args.batch, 480, 640, 3), generator=gb, dtype=torch.uint8).to(dev)
        cur.update(x=bgr, syn=None)
        eb, rawb = timed(model, post, in_flight, args.steps, args.warmup, False)
        t_cv = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            e0.record(); model.cv_preprocess_u8(bgr, io["input_shape"]); e1.record()
            torch.cuda.synchronize(dev)
            t_cv.append(e0.elapsed_time(e1))
        cv_ms = sorted(t_cv)[3]
        cv_bytes = bgr.numel() + args.batch * H * W          # every source byte in, every net-sized byte out
        extras["bgr"] = {"workload": "configs[1]'s batch entering as cv2.imread's frames, uint8 [256,480,640,3] BGR: cvtColor + cv2.resize (any size) on "
                                     "the device (yf_forward_bgr_u8, detect.py:108-127) -> model -> decode -> NMS",
                         "value": round(args.batch * args.steps / eb, 1), "unit": "frames/s", "ms_per_step": round(1e3 * eb / args.steps, 4),
                         "steps": args.steps, "warmup": args.warmup, "in_flight": in_flight,
                         "cv_pre_kernel": {"ms": round(cv_ms, 4), "algorithmic_bytes": cv_bytes, "GBps": round(cv_bytes / (cv_ms * 1e-3) / 1e9, 1),
                                           "hbm_frac": round(cv_bytes / (cv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "bound": "hbm (byte work)"}}
        del bgr
        cur.clear(); cur.update(saved)
        # (c) one frame at a time, the reference's actual calling pattern (detect.py:146-171: model(img), then the post-process, each
        #     followed by a host synchronisation): median over 200 frames, eager launches; and the same pass replayed as a HIP graph
        #     (single lane, small head in line: a graph without parallel branches)
        mb, pb = make(args.dtype, 1, 0)
        x1 = saved["x"][:1].contiguous()
        med = lambda v: sorted(v)[len(v) // 2]

        def batch1_record():
            tm, tp = [], []
            with torch.no_grad():
                for i in range(220):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    pred = mb(x1)
                    torch.cuda.synchronize(dev)
                    t1 = time.perf_counter()
                    pb.detect_raw(pred, kmax=args.kmax)
                    torch.cuda.synchronize(dev)
                    t2 = time.perf_counter()
                    if i >= 20:
                        tm.append(1e3 * (t1 - t0)); tp.append(1e3 * (t2 - t1))
            rec = {"eager": {"model_ms": round(med(tm), 4), "post_ms": round(med(tp), 4), "total_ms": round(med([a + b for a, b in zip(tm, tp)]), 4)}}
            try:
                with torch.no_grad():
                    sg = torch.cuda.Stream()
                    sg.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(sg):
                        pb.detect_raw_from_input(x1, kmax=args.kmax)
                    torch.cuda.current_stream().wait_stream(sg)
                    gph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gph):
                        gout = pb.detect_raw_from_input(x1, kmax=args.kmax)
                    tg = []
                    for i in range(220):
                        torch.cuda.synchronize(dev)
                        t0 = time.perf_counter()
                        gph.replay()
                        torch.cuda.synchronize(dev)
                        if i >= 20:
                            tg.append(1e3 * (time.perf_counter() - t0))
                    ref1 = pb.detect_raw_from_input(x1, kmax=args.kmax)
                    gph.replay()
                    torch.cuda.synchronize(dev)
                    rec["graph_replay"] = {"total_ms": round(med(tg), 4), "what": "yf_detect (model + decode + NMS) captured once, one hipGraphLaunch per frame",
                                           "identical_to_eager": bool(torch.equal(gout["counts"], ref1["counts"]) and torch.equal(gout["head_large"], ref1["head_large"]))}
                    del gph
            except Exception as ex:     # evidence, not product
                rec["graph_replay"] = {"error": repr(ex)[:200]}
            return rec

        # default: at <= 9 frames the stride-32 chain and the small head run as split-sum launches (DESIGN.md section 4 "Small batches"); since round 6
        # they carry the bits of the large-batch kernels, so a frame's logits never depend on the batch size.  Beside it: the one-workgroup launches.
        b1 = {"workload": "one 320x256 frame per call, model then post-process, host-synchronised after each (detect.py:146-171), 200 frames"}
        b1.update(batch1_record())
        with torch.no_grad():
            h_split = [t.clone() for t in mb(x1)]
            h_big = model(saved["x"])
            mb.split_sums = False
            h_one = [t.clone() for t in mb(x1)]
        b1["bits_of_the_batch_256_pass"] = bool(torch.equal(h_split[0], h_big[0][:1]) and torch.equal(h_split[1], h_big[1][:1]))
        b1["split_sums_off"] = batch1_record()
        b1["split_sums_off"]["what"] = "yf_set_split_sums(0): the stride-32 chain and the small head as one workgroup per frame (the same bits)"
        b1["split_sums_off"]["bits_of_the_default"] = bool(torch.equal(h_one[0], h_split[0]) and torch.equal(h_one[1], h_split[1]))
        mb.split_sums = True
        extras["batch1"] = b1
        del mb, pb
        # (d) the N > 1 code path at one rank: a 1-rank RCCL group and the all-gather of every step's packed records -- its frames/s
        #     beside the plain figure is the per-step cost of the exchange machinery (record block + collective launch + wait)
        try:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if "MASTER_ADDR" not in os.environ:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
            dist.init_process_group("nccl", device_id=dev)
            ex_l, pl_l = [], []
            for _ in range(3):      # interleaved, three rounds: consecutive runs of the same loop differ by ~1 %, the cost looked for is of that size
                e_, ex_raw = timed(model, post, in_flight, args.steps, args.warmup, True)
                ex_l.append(e_)
                pl_l.append(timed(model, post, in_flight, args.steps, args.warmup, False)[0])
            ex_e, pl_e = sum(ex_l) / 3, sum(pl_l) / 3
            gathered = cur.get("gathered")
            extras["exchange_rehearsal"] = {
                "workload": "configs[1] with a 1-rank RCCL group: every step's packed record block (yf_decode_nms_packed) all-gathered "
                            "(all_gather_into_tensor), waits pipelined like N > 1",
                "value": round(args.batch * args.steps / ex_e, 1), "unit": "frames/s", "ms_per_step": round(1e3 * ex_e / args.steps, 4),
                "plain_value_same_loop": round(args.batch * args.steps / pl_e, 1), "plain_ms_per_step": round(1e3 * pl_e / args.steps, 4),
                "exchange_overhead_ms_per_step": round(1e3 * (ex_e - pl_e) / args.steps, 4),
                "exchange_overhead_frac_of_step": round((ex_e - pl_e) / pl_e, 4),
                "record_bytes_per_step": int(args.batch * (1 + 8 * args.kmax) * 4),
                "gathered_equals_local": bool(gathered is not None and same_detections(ex_raw, gathered))}
            dist.destroy_process_group()
        except Exception as ex:
            extras["exchange_rehearsal"] = {"error": repr(ex)[:200]}
        cur.clear(); cur.update(saved)

    # The other single-GPU configurations BASELINE.json names, measured in the SAME process after the headline (never `value`):
    #   configs[2]  640x512 batch 128 on the fp16 MFMA path: `f16x3` (split operands: the variant that meets the stated 2e-2 on logits)
    #               and `f16` (fp16 storage, single operands: the throughput mode, 8.5e-2) beside it, each checked against fp32 here;
    #   configs[4]  its per-GPU share: 640x512, 64 frames, dense synthetic head logits, kmax 1024.
    extra_configs = None
    if world == 1 and not args.no_configs and args.res == 256 and args.dtype == "f32" and not args.dense and args.frames == "noise":
        extra_configs = []
        saved = dict(cur)

        def dominant(m, pmc=None):
            """Roofline object of the launch that takes longest.  pmc = (traffic, issue, note) of live_pmc() for this workload: then the
            bound is counter-backed (launch_roofline) and `traffic` / `hbm_frac` are this run's counter bytes; `pass_binding` sums the
            shares over all launches of the pass -- what binds the configuration as a whole."""
            ops_ = m.profile(cur["x"], reps=3, launch_repeats=args.launch_repeats)
            tr, iss, note = pmc if pmc else ({}, {}, None)
            for o in ops_:
                o["roof"] = launch_roofline(o, m.precision, tr.get(o["name"]), iss.get(o["name"]))
            d = max(ops_, key=lambda o: o["ms"])
            r = d["roof"]
            unit = "GB/s" if r["bound"] == "hbm" else "TFLOP/s"
            out_ = {"kernel": d["name"] if len(d["name"]) < 48 else d["name"][:20] + ".." + d["name"][-24:], "launch_ms": round(d["ms"], 4),
                    "bound": r["bound"], "achieved": round(r["hbm_gbs"] if r["bound"] == "hbm" else r["achieved_tf"], 2),
                    "peak": HBM_PEAK_GBS if r["bound"] == "hbm" else r["peak_tf"], "unit": unit, "frac": round(r["frac"], 4),
                    "traffic": tr.get(d["name"]), "hbm_frac": None if r["hbm_frac"] is None else round(r["hbm_frac"], 4),
                    "launches": len(ops_), "sum_of_launch_ms": round(sum(o["ms"] for o in ops_), 4)}
            if "issued_frac" in r:     # fp16 engines: `frac` counts useful flops; the split-operand form issues three MFMAs per product
                out_["issued_frac"] = round(r["issued_frac"], 4)
            if "counters" in r:
                out_["counters"] = r["counters"]
                tot = sum(o["ms"] for o in ops_)
                keys = ("valu_frac", "mfma_frac", "hbm_frac", "lds_frac", "parked_frac")
                out_["pass_binding"] = {k: round(sum(o["roof"]["counters"][k] * o["ms"] for o in ops_) / tot, 3) for k in keys}
                out_["pass_binding"]["hbm_bytes"] = int(sum(tr.get(o["name"], 0) for o in ops_))
                out_["pass_binding"]["launches_by_bound"] = {b: sum(1 for o in ops_ if o["roof"]["bound"] == b) for b in sorted({o["roof"]["bound"] for o in ops_})}
                out_["bound_source"] = ("counters measured in this run (rocprofv3 --pmc child passes: FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU, "
                                        "SQ_VALU_MFMA_BUSY_CYCLES, SQ_WAIT_ANY / SQ_WAVE_CYCLES)")
            elif note:
                out_["traffic_note"] = note
            return out_

        io5, x5, _ = workload(512, 128, "noise", False)
        cur.update(x=x5, syn=None, kmax=64, n_total=128)
        ref_m, ref_p = make("f32", 2, 1, io5, 512)
        with torch.no_grad():
            ref_heads = ref_m(x5)
            ref_raw = ref_p.detect_raw(ref_heads, kmax=64)
        for dt in ("f16x3", "f16"):
            mc, pc = make(dt, 1, 0, io5, 512)
            ec, rawc = timed(mc, pc, 2, args.steps, args.warmup, False)
            with torch.no_grad():
                hc = mc(x5)
            extra_configs.append({
                "config": "BASELINE.json configs[2]: YOLO-Fastest 640x512 batch=128, fp16 MFMA path" +
                          (" (f16x3: fp32 storage, split fp16 operands -- the variant that meets 2e-2 on logits)" if dt == "f16x3" else
                           " (f16: fp16 storage, single fp16 operands -- throughput mode, does not meet 2e-2)"),
                "dtype": dt, "value": round(128 * args.steps / ec, 1), "unit": "frames/s", "ms_per_step": round(1e3 * ec / args.steps, 4),
                "steps": args.steps, "warmup": args.warmup, "in_flight": 2, "lanes": 1,
                "max_abs_logit_diff_vs_f32_on_this_batch": round(max(float((ref_heads[0] - hc[0]).abs().max()), float((ref_heads[1] - hc[1]).abs().max())), 6),
                "detections_identical_to_f32_on_this_batch": bool(same_detections(ref_raw, rawc)),
                "roofline": dominant(mc, cfg2_pmc.get(dt))})
            del mc, pc
        del ref_m, ref_p
        io5, x5, syn5 = workload(512, 64, "noise", True)
        cur.update(x=x5, syn=syn5, kmax=1024, n_total=64)
        for dt in ("f16x3", "f32"):
            mc, pc = make(dt, 1, 0, io5, 512)
            ec, rawc = timed(mc, pc, 2, args.steps, args.warmup, False)
            again = pc.detect_raw(forward(mc), kmax=1024)           # the same records from a second, untimed evaluation
            post64 = post_mean_ms(mc, pc, 1024)                     # mean of 10 warmed repetitions on the same logits
            cnt = rawc["counts"].cpu().numpy()
            cand = float(((syn5[0][:, 4::8] > 0).sum() + (syn5[1][:, 4::8] > 0).sum()).item()) / 64   # sigmoid(t) > 0.5 <=> t > 0
            extra_configs.append({
                "config": "BASELINE.json configs[4], per-GPU share: YOLO-Fastest 640x512, 64 frames, dense synthetic head logits "
                          "(SURVEY.md 8(d).5), kmax 1024" + (" -- fp32 beside it" if dt == "f32" else ""),
                "dtype": dt, "value": round(64 * args.steps / ec, 1), "unit": "frames/s", "ms_per_step": round(1e3 * ec / args.steps, 4),
                "steps": args.steps, "warmup": args.warmup, "in_flight": 2, "lanes": 1,
                "candidates_per_frame": round(cand, 1), "survivors_per_frame": round(float(np.clip(cnt, 0, None).mean()), 1),
                "frames_over_kmax_or_failed": int((cnt < 0).sum()), "post_ms_per_64_frames": round(post64, 4),
                "detections_identical_on_re_evaluation": bool(same_detections(rawc, again)),
                "roofline": dominant(mc)})
            del mc, pc
        cur.clear(); cur.update(saved)

    if rank == 0 and args.dump_records:
        torch.cuda.synchronize(dev)
        rec = cur.get("gathered") if multi else raw
        np.savez(args.dump_records, world_size=world, **{k: rec[k].cpu().numpy() for k in REC})

    if rank == 0:
        counts = raw["counts"].cpu().numpy()
        fps = n_total * args.steps / elapsed
        # per-launch timing, HIP events on the launch stream around every kernel of one forward pass (mean of 5 passes).  An event pair
        # around ONE launch also times the event packets and the dispatch gap behind them -- 5-7 us per launch on some hosts, which is why
        # rocprofv3's kernel durations of the same pass are shorter (profiles/r04_lanes1_kernel_stats.csv: 936 us against 1067 us over the
        # 21 launches) -- so every launch is issued args.launch_repeats times back to back between its two events and the time divided
        # (yf_set_profile_repeats; a launch writes its whole output from inputs it does not modify).  The one-launch-per-event-pair sum
        # stays in the line beside it.
        ops = model.profile(x, reps=5, launch_repeats=args.launch_repeats)
        chain_ms_pairs = sum(o["ms"] for o in model.profile(x, reps=5, launch_repeats=1))
        if args.dump_ops:
            with open(args.dump_ops, "w") as f:
                json.dump([{"name": o["name"], "ms": o["ms"], "algorithmic_bytes": o["algorithmic_bytes"], "mfma_flops": o["mfma_flops"],
                            "valu_flops": o["valu_flops"], "dispatches": o.get("dispatches", 1)} for o in ops], f)
        chain_ms = sum(o["ms"] for o in ops)
        # measured HBM bytes per launch (PMC passes of tools/refresh_profiles.sh), valid for this build and this workload only
        traffic, traffic_note, traffic_source = {}, None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        wl = {"res": args.res, "batch": args.batch, "dtype": args.dtype}
        if live:
            traffic = live
            traffic_source = "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes (bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024)"
        elif os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("source_hash") != source_hash():
                traffic_note = f"profiles/pmc_traffic.json was taken at source hash {tj.get('source_hash')}, this build is {source_hash()}"
            elif tj.get("workload") != wl:
                traffic_note = f"profiles/pmc_traffic.json is for {tj.get('workload')}, this run is {wl}"
            else:
                traffic = {k: v.get("hbm_bytes_per_launch") for k, v in tj.get("kernels", {}).items()}
                traffic_source = "profiles/pmc_traffic.json (same source hash and workload)"
            if live_note and live_note != "not requested":
                traffic_note = (traffic_note + "; " if traffic_note else "") + "live PMC: " + live_note
        else:
            traffic_note = "profiles/pmc_traffic.json not found" + ("; live PMC: " + live_note if live_note else "")
        for o in ops:
            o["roof"] = launch_roofline(o, args.dtype, traffic.get(o["name"]), live_issue.get(o["name"]) if live else None)
        dom = max(ops, key=lambda o: o["ms"])
        dr = dom["roof"]
        bytes_sum = sum(o["algorithmic_bytes"] for o in ops) / args.batch
        bpf = BYTES_PER_FRAME[args.res] // (2 if args.dtype == "f16" else 1)   # f16x3 stores fp32
        total_traffic = sum(traffic.get(o["name"], 0) or 0 for o in ops) if traffic else None
        chain_floor = sum(o["roof"]["compute_frac"] * o["ms"] * 1e-3 for o in ops)   # sum of the launches' floors
        roofline = {"bound": dr["bound"], "kernel": dom["name"], "launch_ms": round(dom["ms"], 4),
                    "share_of_forward": round(dom["ms"] / chain_ms, 4), "traffic": traffic.get(dom["name"]),
                    "compute": {"achieved": round(dr["achieved_tf"], 2), "peak": dr["peak_tf"], "unit": "TFLOP/s",
                                "frac": round(dr["compute_frac"], 4),
                                "pipe": ("fp16 MFMA" if args.dtype != "f32" and dr["on_mfma"] else
                                         "fp32 issue: MFMA f32 + VALU share 64 FLOP/clk/SIMD" if args.dtype == "f32" else "fp32 VALU"),
                                "mfma_flops_per_launch": int(dom["mfma_flops"]), "valu_flops_per_launch": int(dom["valu_flops"])},
                    "hbm": {"achieved": None if dr["hbm_gbs"] is None else round(dr["hbm_gbs"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": None if dr["hbm_frac"] is None else round(dr["hbm_frac"], 4)},
                    # NOT a roofline: what the layers of this launch would move if every conv read its input and wrote its
                    # output through HBM, over the fused launch's time; > 1.0 by construction for a fused plan
                    "layer_granular_equiv": {"algorithmic_bytes_per_launch": int(dom["algorithmic_bytes"]),
                                             "GBps_equiv": round(dom["algorithmic_bytes"] / (dom["ms"] * 1e-3) / 1e9, 1),
                                             "over_hbm_peak": round(dom["algorithmic_bytes"] / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
        if dr["bound"] == "hbm":
            roofline.update(achieved=roofline["hbm"]["achieved"], peak=HBM_PEAK_GBS, unit="GB/s", frac=roofline["hbm"]["frac"])
        else:
            roofline.update(achieved=roofline["compute"]["achieved"], peak=dr["peak_tf"], unit="TFLOP/s", frac=roofline["compute"]["frac"])
        if "issued_frac" in dr:
            roofline["issued_frac"] = round(dr["issued_frac"], 4)
        # The three pass-level figures SURVEY.md 8(d) asks a block-fused plan to publish next to the dominant launch's roofline, over the
        # HEADLINE's own step time (whole job, all launches + post-process, two batches in flight): useful flops over the fp32 issue peak
        # (fp16 engines: over the peak of the pipe each launch's flops run on, summed as launch floors), counter-measured HBM bytes over
        # 8 TB/s, and the layer-granular figure (what an unfused network would move; > 1 by construction, NOT a roofline fraction).
        step_s = elapsed / args.steps
        per_rank_frames = n_total / world
        pass_flops = FLOPS_PER_FRAME[args.res] * per_rank_frames
        roofline["pass_compute_frac"] = round((pass_flops / (FP32_PEAK_TF * 1e12) if args.dtype == "f32" else chain_floor) / step_s, 4)
        roofline["pass_hbm_frac"] = None if not total_traffic else round(total_traffic * (per_rank_frames / args.batch) / step_s / 1e9 / HBM_PEAK_GBS, 4)
        roofline["pass_hbm_bytes_per_frame"] = None if not total_traffic else int(total_traffic / args.batch)
        roofline["layer_granular_equiv_pass"] = round(per_rank_frames * bpf / step_s / 1e9 / HBM_PEAK_GBS, 4)
        roofline["pass_note"] = ("pass_* = whole pass over the headline's step time: useful flops / fp32 issue peak, PMC bytes / 8 TB/s; "
                                 "layer_granular_equiv_pass = SURVEY.md 8(d) bytes per frame x frames / step / 8 TB/s (exceeds 1 for a fused plan)")
        if "counters" in dr:      # what the bound was decided on: the launch's issue-side counters of this run (launch_roofline)
            roofline["counters"] = dr["counters"]
            roofline["bound_source"] = ("counters measured in this run (rocprofv3 --pmc child passes: FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU, "
                                        "SQ_VALU_MFMA_BUSY_CYCLES, SQ_WAIT_ANY / SQ_WAVE_CYCLES); fp32 MFMA and VALU share the issue slots (fp32_issue_frac)")
        if traffic_note:
            roofline["traffic_note"] = traffic_note
        if traffic_source:
            roofline["traffic_source"] = traffic_source
        out = {
            "metric": "frames/sec end-to-end (model forward + decode + per-class NMS), 320x256 batch=256 per GPU"
                      if args.res == 256 else "frames/sec end-to-end, 640x512",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "ms_per_frame": round(1e3 * elapsed / args.steps / n_total, 6),
            # `value` is region 0; the same K-step region repeated in the same loop (no re-warm-up, same streams): the spread of the headline
            "repeat_values": [round(n_total * args.steps / t, 1) for t in region_s],
            "settle_steps": args.settle, "warmup_effective": args.warmup + args.settle,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic" if args.frames == "noise" else "the reference's 20 bundled frames tiled to the batch",
            "config": {"workload": f"YOLO-Fastest {W}x{H} batch={args.batch} fp32 per GPU, synthetic uniform-u8 frames "
                                   f"(BASELINE.json configs[1])" if args.res == 256 and args.batch == 256 and args.dtype == "f32" and args.frames == "noise" else
                                   f"YOLO-Fastest {W}x{H} batch={args.batch} {args.dtype} per GPU, {args.frames} frames"
                                   + (", dense synthetic head logits (SURVEY.md 8(d) config 5)" if args.dense else ""),
                       "global_batch": n_total, "weights": wname, "kmax": args.kmax, "chunk": args.chunk, "in_flight": in_flight, "lanes": lanes, "branches": branches,
                       "world_size": world, "cpu_affinity": affinity,
                       "parallelism": (f"dp{world} REHEARSAL: {world} ranks share ONE GPU, records exchanged over gloo -- not a performance figure" if args.share_gpu and world > 1
                                       else f"dp{world} (frames sharded, one RCCL all-gather of box records)" if world > 1 else "single GPU"),
                       "survivors_per_frame_mean": round(float(np.clip(counts, 0, None).mean()), 3)},
            "roofline": roofline,
            # the whole forward pass (all launches)
            "forward_chain": {"launches": len(ops), "forward_ms": round(fwd_ms, 4), "post_ms": round(post_ms, 4),
                              "sum_of_launch_ms_single_stream": round(chain_ms, 4),
                              "launch_timing": f"HIP events around {args.launch_repeats} back-to-back launches of each kernel, divided by {args.launch_repeats} (agrees with "
                                               "rocprofv3's kernel durations); one launch per event pair also times the event packets and the dispatch gap",
                              "sum_of_launch_ms_one_launch_per_event_pair": round(chain_ms_pairs, 4),
                              "compute_frac": round(chain_floor / (fwd_ms * 1e-3), 4),
                              "hbm_traffic_bytes": total_traffic,
                              "hbm_GBps": None if not total_traffic else round(total_traffic / (fwd_ms * 1e-3) / 1e9, 1),
                              "hbm_frac": None if not total_traffic else round(total_traffic / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "layer_granular_equiv": {"algorithmic_bytes_per_frame": bpf,
                                                       "algorithmic_bytes_per_frame_sum_over_launches": int(bytes_sum),
                                                       "GBps_equiv": round(args.batch * bpf / (fwd_ms * 1e-3) / 1e9, 1),
                                                       "over_hbm_peak": round(args.batch * bpf / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                              "per_launch": [{"name": o["name"] if len(o["name"]) < 48 else o["name"][:20] + ".." + o["name"][-24:],
                                              "ms": round(o["ms"], 4), "dtype": o["kernel_dtype"], "bound": o["roof"]["bound"],
                                              "frac": round(o["roof"]["frac"], 3),
                                              "hbm_GBps": None if o["roof"]["hbm_gbs"] is None else round(o["roof"]["hbm_gbs"]),
                                              # issue-side counter shares of the launch's time: VALU issue, MFMA busy, waves parked
                                              "issue": None if "counters" not in o["roof"] else [o["roof"]["counters"]["valu_frac"],
                                                                                                  o["roof"]["counters"]["mfma_frac"],
                                                                                                  o["roof"]["counters"]["parked_frac"]]}
                                             for o in ops]},
        }
        if single is not None:
            out["one_batch_in_flight"] = single
        if other is not None:
            out["two_batches_in_flight"] = other
        if variant is not None:
            out["variants"] = [variant]
        if extra_configs is not None:
            out["configs"] = extra_configs
        if extras is not None:
            out.update(extras)      # fixtures, u8, batch1, exchange_rehearsal
        if world == 1 and not args.no_train and args.res == 256 and args.dtype == "f32" and not args.dense and args.frames == "noise":
            del model, post
            torch.cuda.empty_cache()
            out["training"] = train_records(dev, train_pmc)        # never `value`
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
