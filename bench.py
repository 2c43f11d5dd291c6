#!/usr/bin/env python3
"""Headline benchmark: complex MSamples/s demodulated (uint8 IQ -> packed bits -> preamble
matches -> 10-byte packets with RSSI/SNR) on synthetic Davis streams, per BASELINE.json.

    python bench.py --gpus N --steps K --warmup W

One process per GPU: for N > 1 either under a launcher (torch.distributed.run sets RANK / WORLD_SIZE) or plainly as
`python bench.py --gpus N`, which starts the N ranks itself (launch_ranks).  Streams are
independent, so each rank demodulates its own shard and no collective touches the data
path (weak scaling: 4096 streams per GPU, BASELINE.json configs[3] / configs[4]).  A
step = one full pass over the rank's resident batch: rd_batch_run (all kernels) followed by
rd_batch_results (device->host copy of the packets, per-call ordering and dedupe).
Two
resident copies of the batch are demodulated alternately so that the host part of step i
(results) overlaps the GPU part of step i+1 - the double-buffered shape of a receiver that
demodulates one capture while the next one arrives; every step still does all of its work
inside the timed region.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (fused demod):
algorithmic 2 B per complex sample / its mean HIP-event duration on the launch stream.
`cpu_baseline` times the reference's algorithm (oracle/: per-block NumPy, vectorised NumPy, C port) on
this box's host cores on a bounded sample of the same streams.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# before anything initialises HIP (torch does): small device->host copies on the SDMA engines, not on a
# blit kernel that competes with the demod kernel for the CUs (see rtldavis_amd/__init__.py)
os.environ.setdefault("GPU_FORCE_BLIT_COPY_SIZE", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
UNIQUE = 64            # unique synthetic streams, tiled to fill the batch


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle", type=float, default=0.05,
                    help="seconds of untimed steps in front of the W warmup steps (reported as `settle`): the chip "
                         "needs ~30 ms of continuous load after the idle upload phase before its clocks settle - "
                         "a 20-step region started cold runs the demod kernel at 0.57 ms, warm at 0.52 "
                         "(profiles/r02_clock_settle.txt); 0 switches it off")
    ap.add_argument("--streams", type=int, default=4096, help="streams per GPU")
    ap.add_argument("--blocks", type=int, default=33, help="8192-sample blocks per stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--live-traffic", type=int, default=1,
                    help="1 (default, N = 1 only): after the timed region, measure roofline.traffic on THIS box - two child "
                         "runs of this script under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE: separate passes, counters "
                         "only), ~30 s; 0 or any failure: the committed profiles/rNN_traffic.json whose kernel stamp matches")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--sustain", type=float, default=6.0,
                    help="seconds of back-to-back steps after the timed region (reported separately, never `value`; "
                         "0 = off).  Longer than a 5 s utilisation sampler's period, so that one cannot miss the GPU work")
    ap.add_argument("--two-streams", action="store_true", help="one HIP stream per resident batch")
    ap.add_argument("--step-times", action="store_true", help="diagnostic: print the slowest steps' host times to stderr")
    ap.add_argument("--resident", type=int, default=2,
                    help="resident copies of the batch demodulated round-robin (>= 2): runs queued ahead of the host; "
                         "3 and 4 measured no better than 2 (profiles/r02_readback_sdma.txt)")
    ap.add_argument("--pipelined", type=int, default=0,
                    help="1: rd_batch_set_pipelined on every resident batch - a run's completion rides on the next run's "
                         "demod kernel instead of an event on its own last kernel (no idle gap between runs; use with "
                         "--resident 3 so that the host stays a whole step ahead)")
    ap.add_argument("--stage-times", action="store_true", help="time every kernel stage (adds events)")
    ap.add_argument("--wideband", action="store_true",
                    help="BASELINE configs[2] instead of the headline workload: 51 hop channels out of one "
                         "synthetic 26.88 MS/s capture (channelizer + demodulator), N = 1 only")
    ap.add_argument("--host-fed", action="store_true",
                    help="also measure the path fed from (pinned) host memory: upload(i+1) on a copy stream beside run(i). "
                         "Reported as a separate `host_fed` object - PCIe-bound by construction, never `value`")
    ap.add_argument("--dist-backend", choices=("auto", "nccl", "gloo"), default="auto",
                    help="N > 1: backend of the barrier / max-over-ranks time (no data-path collective exists). auto = nccl (RCCL), "
                         "falling back to gloo in the same process when its initialisation fails")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="N > 1 dry run on a one-GPU box: all ranks on cuda:0, gloo barrier (not a measurement)")
    return ap.parse_args()


def cpu_baseline(uniq: np.ndarray, budget_s: float = 10.0):
    """The reference's algorithm on this box's host cores, bounded sample, three legs:
    (i) per-block NumPy exactly as dsp.Demodulator.demodulate runs it (dsp.py:139-169, incl. its
    per-sample quantize loop, dsp.py:93-98) - comparable with the 1.77 MS/s measured for the real
    reference in the survey container; (ii) vectorised NumPy, whole stream at once; (iii) the C port
    (oracle/dsp_oracle.c) built for the highest x86-64 level the host supports, multi-threaded over
    streams.  `value` is leg (iii); all three are in `legs`."""
    from oracle import c_oracle as CO, dsp_oracle as O
    legs = {}
    ocfg = O.production_config()
    B = ocfg.block_size
    # (i) one core, block by block
    nblk = uniq.shape[1] // (2 * B)
    t0 = time.perf_counter()
    done = 0
    for i in range(uniq.shape[0]):
        dem = O.OracleDemodulator(ocfg, per_sample_quantize=True)
        for blk in range(nblk):
            dem.demodulate(uniq[i, 2 * B * blk: 2 * B * (blk + 1)])
            done += B
        if time.perf_counter() - t0 > 3.0:
            break
    dt = time.perf_counter() - t0
    legs["numpy_per_block"] = {"value": round(done / dt / 1e6, 3), "unit": "MS/s", "cores": 1,
                               "sample": f"{done // B} blocks of {done // B // nblk} streams ({done / 1e6:.2f} MS, {dt:.1f} s), "
                                         "OracleDemodulator with the reference's per-sample quantize loop"}
    # (ii) one core, whole streams at once
    t0 = time.perf_counter()
    done = 0
    for i in range(uniq.shape[0]):
        f, d, bits = O.demod_stream_oneshot(uniq[i])
        O.calls_from_oneshot(f, bits, ocfg)
        done += uniq.shape[1] // 2
        if time.perf_counter() - t0 > 3.0:
            break
    dt = time.perf_counter() - t0
    legs["numpy_vectorised"] = {"value": round(done / dt / 1e6, 2), "unit": "MS/s", "cores": 1,
                                "sample": f"{done // (uniq.shape[1] // 2)} whole streams ({done / 1e6:.1f} MS, {dt:.1f} s), "
                                          "demod_stream_oneshot + calls_from_oneshot"}
    # (iii) C port, best ISA level, all cores
    CO.build()
    level = CO.use_isa_level(4)
    cfg = CO.make_cfg()
    ncpu = os.cpu_count() or 1
    work = np.tile(uniq, (4, 1))
    samples = work.shape[0] * work.shape[1] // 2
    best = None
    for th in sorted({1, min(ncpu, 8), min(ncpu, 32), min(ncpu, 64), min(ncpu, 128), ncpu}):
        CO.demod_batch(work[: max(1, min(work.shape[0], th))], cfg, th)  # warm the thread pool
        t0 = time.perf_counter()
        CO.demod_batch(work, cfg, th)
        dt = time.perf_counter() - t0
        if th == 1:
            legs["c_port_1_thread"] = {"value": round(samples / dt / 1e6, 2), "unit": "MS/s", "cores": 1,
                                       "sample": f"{work.shape[0]} streams, one pass, x86-64-v{level}"}
        if best is None or samples / dt > best[0]:
            best = (samples / dt, th, dt)
    _, th, dt = best
    reps = int(max(1, min(200, budget_s / max(dt, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        CO.demod_batch(work, cfg, th)
    dt = time.perf_counter() - t0
    CO.use_isa_level(2)
    legs["c_port"] = {"value": round(samples * reps / dt / 1e6, 2), "unit": "MS/s", "cores": th,
                      "sample": f"x86-64-v{level} build, best of 1..{ncpu} threads"}
    return {"value": legs["c_port"]["value"], "unit": "MS/s", "cores": th, "kind": "port",
            "sample": f"{work.shape[0]} streams ({uniq.shape[0]} unique) x {work.shape[1] // 2} samples, {reps} passes "
                      f"({samples * reps / 1e6:.0f} MS, {dt:.1f} s wall), C port built -march=x86-64-v{level}, best of "
                      f"1..{ncpu} threads on {ncpu} host cpus ({_cpu_model()})",
            "legs": legs}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown cpu"


class ChipSampler:
    """Package power and shader clock as the chip reports them (rocm-smi, an ordinary user may read them) while the
    sustained leg runs: a thread of its own, a sample per second from second 1.5 on, nothing if the tool is missing or
    slow.  DESIGN 5.1: the run sits on the package's power cap - a number the reader should not have to take on trust."""

    def __init__(self, device: int):
        import threading
        self.device, self.samples, self.cap, self._stop = device, [], None, threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _smi(self, *flags):
        import subprocess
        return subprocess.run(["rocm-smi", *flags, "-d", str(self.device)], capture_output=True, text=True, timeout=5).stdout

    def _run(self):
        import re
        try:
            if self._stop.wait(1.5):
                return
            while not self._stop.is_set() and len(self.samples) < 8:
                out = self._smi("--showpower", "--showclocks")
                pw = re.search(r"Power \(W\): ([0-9.]+)", out)
                ck = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
                if pw:
                    self.samples.append((float(pw.group(1)), int(ck.group(1)) if ck else None))
                if self._stop.wait(1.0):
                    break
            m = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)", self._smi("--showmaxpower"))
            self.cap = float(m.group(1)) if m else None
        except Exception:
            pass

    def stop(self) -> dict:
        self._stop.set()
        self._t.join(timeout=8)
        if not self.samples:
            return {}
        pw = [p for p, _ in self.samples]
        ck = [c for _, c in self.samples if c is not None]
        out = {"package_power_W": [min(pw), max(pw)], "power_samples": len(pw)}
        if ck:
            out["sclk_MHz"] = [min(ck), max(ck)]
        if self.cap:
            out["power_cap_W"] = self.cap
        return out


def wideband(args):
    """configs[2]: one step = channelize one second of wideband capture into the batch demodulator's
    input, demodulate the 51 channels, fetch the packets.  Parity here is end to end (every injected
    packet must come back); the channelizer itself has no reference counterpart."""
    from rtldavis_amd import batch, channelizer as CZ, dsp, synth
    nb = args.blocks
    n_out = nb * 8192
    cz = CZ.Channelizer()
    raw, info = synth.synth_wideband(range(100, 100 + cz.n_channels), [f - CZ.DEFAULT_CENTRE_HZ for f in CZ.US_CHANNELS_HZ],
                                     n_out, amplitude=0.05)
    cz.upload(raw)
    cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
    bd = batch.BatchDemodulator(cfg, cz.n_channels, nb)

    def step():
        cz.run_into(bd)
        bd.run()
        return bd.results()

    for _ in range(max(1, args.warmup)):
        recs = step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        recs = step()
    dt = (time.perf_counter() - t0) / args.steps
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")  # (the library's own HIP runtime: already loaded)
    hip.hipDeviceSynchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        cz.run_into(bd)
    if hip.hipDeviceSynchronize() != 0:  # (the launches above are on the null stream)
        raise SystemExit("bench.py --wideband: hipDeviceSynchronize failed")
    dt_c = (time.perf_counter() - t1) / args.steps
    ok = sum(payload in [r["data"][: int(r["nbytes"])].tobytes().hex() for r in recs if int(r["stream"]) == c]
             for c, (payload, _s) in enumerate(info))
    if ok != len(info):
        raise SystemExit(f"bench.py --wideband: only {ok} of {len(info)} injected packets recovered - result invalid")
    flops = 8.0 * cz.n_channels * n_out * cz.taps.size
    print(json.dumps({
        "metric": "wideband complex MSamples/s channelized into 51 hop channels and demodulated",
        "value": round(n_out * cz.decim / dt / 1e6, 1), "unit": "MS/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * dt, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16 (channelizer, two tap digits, fp32 accumulate) + f32 (demod)", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[2]: 51 US hop channels from one {n_out * cz.decim / 1e6:.1f} M-sample "
                               f"uint8 IQ capture at 26.88 MS/s ({n_out / CZ.OUT_RATE:.2f} s of air), 1 MI355X",
                   "channels": cz.n_channels, "taps": int(cz.taps.size), "decimation": cz.decim},
        "roofline": {"bound": "mfma", "achieved": round(flops / dt_c / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": round(flops / dt_c / 1e12 / 2500.0, 4), "traffic": None, "kernel": "k_channelize",
                     "kernel_ms": round(1e3 * dt_c, 4),
                     "note": "useful flops (8 x taps per output); the kernel issues 2x as many f16 MACs (taps split into two digits) "
                             "on 128 rows for 102; kernel_ms is wall time per launch, back to back"},
        "packets_recovered": f"{ok} of {len(info)}", "real_time_factor": round(n_out / CZ.OUT_RATE / dt, 1),
        "parity": "unpinned: rtldavis has no channelizer to compare with; checked against this repo's float64 model "
                  "(<= 1 LSB) and by recovering the injected packets through the pinned demodulator"}), flush=True)


def live_traffic(args, budget_s: float = 150.0):
    """HBM traffic of the demod kernel per launch, measured on this box: FETCH_SIZE and WRITE_SIZE in separate
    `rocprofv3 --pmc` passes (counters only, the program straight after `--`), each a child run of this script with a
    handful of steps; corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x 2 on gfx950, KiB -> bytes).
    Returns (bytes or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this run is itself under a profiler"
    vals = {}
    t_end = time.perf_counter() + budget_s
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="rd_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", c, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-verify", "--sustain", "0", "--settle", "0",
               "--live-traffic", "0", "--streams", str(args.streams), "--blocks", str(args.blocks)]
        env = dict(os.environ)
        env["TMPDIR"] = "/tmp"
        try:
            left = t_end - time.perf_counter()
            if left < 10:
                return None, "time budget spent"
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=left)
            acc = []
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if "k_demod" in r["Kernel_Name"] and r["Counter_Name"] == c:
                            acc.append(float(r["Counter_Value"]))
            if not acc:
                return None, f"no {c} rows from rocprofv3"
            vals[c] = sum(acc) / len(acc)
        except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
            return None, f"{c} pass failed: {type(e).__name__}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024), None


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU through
    torch.distributed.run, rendezvous on 127.0.0.1) and hand back their exit code.  This parent process never
    touches the GPU (no HIP call, no torch import) and never exec()s: the ranks are ordinary children, rank 0's
    JSON line goes straight to our stdout."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:  # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this platform (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    try:
        return subprocess.run(cmd, env=env).returncode
    except KeyboardInterrupt:
        return 130


def main():
    args = parse_args()
    if args.wideband:
        return wideband(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback)")
    # --rehearse-shared-gpu: every rank on cuda:0 with a gloo barrier, to exercise the N > 1 code
    # path on a one-GPU box (RCCL refuses two ranks on one device); never a measurement
    rehearse = args.rehearse_shared_gpu
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist_backend, dist_note = None, None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        want = "gloo" if rehearse else args.dist_backend
        if want in ("auto", "nccl"):
            # The path shards without any exchange: torch.distributed only carries the barrier, the
            # max-over-ranks time and the gathered per-GPU lines.  RCCL is the contract's backend; a run that needs no
            # collective must not die because RCCL cannot come up (IPC mode, a busy fabric), so its failure - at
            # initialisation or at the first barrier - falls back to gloo in this same process (no re-exec).
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
                dist.barrier()
                torch.cuda.synchronize()
                dist_backend = "nccl"
            except Exception as e:  # noqa: BLE001 - whatever RCCL raises
                if want == "nccl":
                    raise
                dist_note = f"nccl failed ({type(e).__name__}: {str(e)[:120]}): gloo"
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                except Exception:  # noqa: BLE001
                    pass
        if dist_backend is None:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist_backend = "gloo"
    on_cpu = dist_backend != "nccl"   # where the tensors of the (two) scalar reductions live

    from rtldavis_amd import _lib, batch, dsp, synth

    _lib.check(_lib.lib().rd_set_device(local_rank))
    cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
    n_streams, n_blocks = args.streams, args.blocks
    n_samples = n_blocks * cfg.block_size

    # synthetic input: UNIQUE streams tiled into distinct HBM addresses.  Every rank holds the same
    # 64 unique streams (seeds 0..63, the ones tests/golden/streams.json covers) so that every rank's
    # output is verified against the reference fixtures, not rank 0's alone.
    nu = min(UNIQUE, n_streams)
    seeds = list(range(nu))
    uniq = synth.synth_streams(seeds, n_samples=n_samples) if n_blocks == synth.BLOCKS_PER_STREAM else \
        np.stack([synth.synth_stream(s, n_samples=max(n_samples, 3 * 8192 + 2000))[: 2 * n_samples] for s in seeds])
    reps = (n_streams + nu - 1) // nu
    host = np.tile(uniq, (reps, 1))[:n_streams]
    bds = [batch.BatchDemodulator(cfg, n_streams, n_blocks) for _ in range(max(2, args.resident))]
    bd = bds[0]
    t_h2d = time.perf_counter()
    bd.upload(host)
    torch.cuda.synchronize()
    t_h2d = time.perf_counter() - t_h2d
    for x in bds[1:]:
        x.upload(host)
    if args.pipelined:
        for x in bds:
            x.set_pipelined(True)
    in_bytes = host.nbytes
    host_pinned = None
    if args.host_fed:   # the same bytes in pinned memory, for the overlapped uploads measured after the timed region
        host_pinned = torch.from_numpy(host).pin_memory()
    del host

    # --two-streams gives each resident batch its own HIP stream; measured: no gain, the demod
    # kernel's persistent grid leaves no room for another batch's kernels to run beside it
    tstreams = [torch.cuda.Stream() for _ in bds] if args.two_streams else [torch.cuda.current_stream()] * len(bds)
    streams = [t.cuda_stream for t in tstreams]
    # Kernel timing (HIP events riding on the dispatches) is switched on right in front of the timed region:
    # the settle and warmup steps run untimed, so there are no event pairs to read back - a read-back of a
    # hundred of them idles the GPU long enough for its clock to drop again.
    timing_level = 2 if args.stage_times else 1  # 1: demod kernel + whole run (an event between two
    # kernels idles the GPU for ~6 us, so the per-stage split is opt-in and comes from rocprofv3)

    R = len(bds)
    step_log = [] if args.step_times else None  # diagnostic: host time of run() and results() per step

    def run_steps(k):
        """k full steps; step i = bds[i % R].run + its results().  R resident copies of the batch are
        demodulated round-robin with R - 1 runs queued ahead, so that the host part of step i (results:
        wait, D2H of the packets, per-call ordering and dedupe, ~0.4 ms) overlaps the GPU part of the
        following steps and the GPU never waits for a launch."""
        recs = None
        for i in range(min(R - 1, k)):
            bds[i % R].run(streams[i % R])
        for i in range(k):
            nxt = i + R - 1
            ta = time.perf_counter() if step_log is not None else 0.0
            if nxt < k:
                bds[nxt % R].run(streams[nxt % R])
            tb = time.perf_counter() if step_log is not None else 0.0
            recs = bds[i % R].results()
            if step_log is not None:
                step_log.append((tb - ta, time.perf_counter() - tb))
        return recs, bds[(k - 1) % R]

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The upload leaves the compute units idle for 0.1-0.3 s and the chip clocks down; it takes ~30 ms of load to
    # come back (profiles/r02_clock_settle.txt).  A step is 0.65 ms, so W = 5 warmup steps end long before that and
    # a short timed region would measure the ramp, not the path.  The settle phase is untimed, disclosed in the
    # output line, and the same steps as everything else; the K timed steps follow the W warmup steps unchanged.
    import gc
    gc.collect()
    gc.disable()  # no collection pause of the interpreter inside the 60 ms timed region (nor an idle GPU right before it)
    settle_steps, settle_s = 0, 0.0
    if args.settle > 0:
        ts = time.perf_counter()
        while time.perf_counter() - ts < args.settle:
            run_steps(4 * R)
            settle_steps += 4 * R
        settle_s = time.perf_counter() - ts
    if args.warmup:
        run_steps(args.warmup)
    sync_all()
    for x in bds:
        x.set_timing(timing_level)  # the timing window holds the K timed steps only
    t0 = time.perf_counter()
    recs, bd = run_steps(args.steps)
    sync_all()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if step_log is not None and rank == 0:
        timed = step_log[-args.steps:]
        worst = sorted(range(len(timed)), key=lambda j: -(timed[j][0] + timed[j][1]))[:6]
        print("step times (ms): median run %.3f results %.3f; slowest: %s" % (
            1e3 * sorted(t[0] for t in timed)[len(timed) // 2], 1e3 * sorted(t[1] for t in timed)[len(timed) // 2],
            ", ".join("#%d run %.2f results %.2f" % (j, 1e3 * timed[j][0], 1e3 * timed[j][1]) for j in worst)), file=sys.stderr)
        del step_log[:]
    tms = [x.timing() for x in bds]  # mean kernel durations (HIP events on the launch stream)
    assert sum(t["runs"] for t in tms) == args.steps
    tm = {k: sum(t[k] * t["runs"] for t in tms) / args.steps for k in tms[0] if k != "runs"}
    own_elapsed = elapsed
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if on_cpu else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- sustained leg (never `value`): the same steps back to back for >= 2 s, so that the clock the chip
    # holds under continuous load shows (the timed region above is a burst of a few tens of ms)
    sustained = None
    if args.sustain > 0 and world == 1:
        k = max(10, int(args.sustain / max(elapsed / args.steps, 1e-4)))
        for x in bds:
            x.timing()
        chip = ChipSampler(local_rank) if args.sustain >= 3 else None   # (what the chip reports meanwhile: rocm-smi, own thread)
        t1 = time.perf_counter()
        run_steps(k)
        sync_all()
        dt_s = time.perf_counter() - t1
        tms2 = [x.timing() for x in bds]
        dm2 = sum(t["demod_ms"] * t["runs"] for t in tms2) / max(1, sum(t["runs"] for t in tms2))
        sustained = {"seconds": round(dt_s, 2), "steps": k, "value": round(n_streams * n_samples * k / dt_s / 1e6, 1),
                     "unit": "MS/s", "ms_per_step": round(1e3 * dt_s / k, 4), "kernel_ms": round(float(dm2), 4),
                     "roofline_frac": round(n_streams * n_samples * 2 / (dm2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if chip is not None:
            sustained.update(chip.stop())

    # ---- host-fed leg (never `value`): every step's input crosses the bus.  upload(i + 1) is issued on a copy stream
    # while run(i) computes (rd_batch_upload_async: the run waits for its upload on the device, the upload for the
    # previous run of its handle); the link decides - 2 bytes per sample over PCIe against 0.55 ms of kernels.
    host_fed = None
    if host_pinned is not None and world == 1:
        hp = host_pinned.numpy().reshape(-1)
        cs = torch.cuda.Stream()
        for x in bds:
            x.set_timing(0)
        sync_all()
        # the link by itself: plain pinned-memory uploads of the same buffer
        t1 = time.perf_counter()
        for _ in range(4):
            bds[0].upload_async(hp, cs.cuda_stream)
        torch.cuda.synchronize()
        link_s = (time.perf_counter() - t1) / 4
        kf = 12
        t1 = time.perf_counter()
        bds[0].upload_async(hp, cs.cuda_stream)
        for i in range(kf):
            if i + 1 < kf:
                bds[(i + 1) % R].upload_async(hp, cs.cuda_stream)   # beside the run below
            bds[i % R].run(streams[i % R])
            frecs = bds[i % R].results()
        torch.cuda.synchronize()
        fed_s = (time.perf_counter() - t1) / kf
        host_fed = {"value": round(n_streams * n_samples / fed_s / 1e6, 1), "unit": "MS/s", "steps": kf,
                    "ms_per_step": round(1e3 * fed_s, 3), "h2d_GBps": round(in_bytes / link_s / 1e9, 2),
                    "frac_of_link": round(link_s / fed_s, 4), "packets_per_step": int(len(frecs)),
                    "note": "inputs in pinned host memory, upload(i+1) on a copy stream beside run(i): PCIe-bound by "
                            "construction (2 B per sample over the link); link = the same uploads alone"}
        recs, bd = bds[(kf - 1) % R].results(), bds[(kf - 1) % R]   # (what the verification below reads)
        for x in bds:
            x.set_timing(timing_level)

    # ---- verification outside the timed region: packets and bit hashes against the fixtures
    verified = None
    if not args.no_verify and n_blocks == synth.BLOCKS_PER_STREAM:
        import hashlib
        with open(os.path.join(ROOT, "tests", "golden", "streams.json")) as fh:
            gold = json.load(fh)
        per = {}
        for r in recs:
            per.setdefault(int(r["stream"]), []).append(
                (int(r["call"]), int(r["index"]), r["data"][: int(r["nbytes"])].tobytes().hex()))
        ok = True
        for s in range(n_streams):
            g = gold[str(s % nu)]
            want = [(int(c), p["index"], p["data"]) for c, ps in sorted(g["calls"].items(), key=lambda kv: int(kv[0]))
                    for p in ps]
            ok &= per.get(s, []) == want
        for s in sorted({0, nu - 1, n_streams // 2, n_streams - 1}):
            ok &= hashlib.sha256(bd.bits(s).tobytes()).hexdigest() == gold[str(s % nu)]["bits_sha256"]
        verified = bool(ok)
        if world > 1:  # every rank must agree
            vt = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device="cpu" if on_cpu else "cuda")
            dist.all_reduce(vt, op=dist.ReduceOp.MIN)
            verified = bool(vt.item() > 0.5)
        own_verified = bool(ok)
        if not verified:
            raise SystemExit(f"bench.py: GPU output differs from the reference fixtures (rank {rank}: "
                             f"{'ok' if ok else 'MISMATCH'}) - result invalid")
    else:
        own_verified = None

    # BASELINE configs[4]: per-GPU MSamples/s next to the aggregate - every rank's own clock, kernel time and
    # verification, gathered on rank 0 (a straggler GPU shows here; MAX(elapsed) alone would hide which one it is)
    per_gpu = None
    if world > 1:
        dm_own = float(tm["demod_ms"])
        mine = {"rank": rank, "device": torch.cuda.get_device_name(local_rank) + f" (cuda:{local_rank})",
                "value": round(n_streams * n_samples * args.steps / own_elapsed / 1e6, 1),
                "ms_per_step": round(1e3 * own_elapsed / args.steps, 4), "kernel_ms": round(dm_own, 4),
                "roofline_frac": round(n_streams * n_samples * 2 / (dm_own * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "verified": own_verified}
        from rtldavis_amd import shard
        per_gpu = shard.gather_rank_lines(mine)

    # HBM traffic of the dominant kernel: measured with PMC counters in separate rocprofv3 runs
    # (tools/profile_round.sh) and committed under profiles/ together with a sha256 over the kernel's INSTRUCTIONS as
    # compiled for the library it was measured on (tools/profile_collect.py::isa_sha256; the build leaves the tree's
    # stamp next to the library); valid for the default workload
    # only.  A file that belongs to another build of the kernel is refused.
    traffic, traffic_note = None, None
    try:
        import glob
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import profile_collect
        # the stamp the build left next to the product library (only that library has one)
        stamp = profile_collect.kernel_isa_stamp() if os.path.samefile(_lib.LIB_PATH, os.path.join(ROOT, "rtldavis_amd", "librtldavis_hip.so")) else ""
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")), reverse=True)
        for path in files:  # the newest round's file that was measured on this very kernel code
            with open(path) as fh:
                tj = json.load(fh)
            w = tj["workload"]
            if not stamp or tj.get("kernel_isa_sha256") != stamp:
                if traffic_note is None:
                    traffic_note = (f"profiles/{os.path.basename(path)} was measured on another build of k_demod_mfma "
                                    f"(commit {tj.get('commit')}): not used; regenerate with tools/profile_round.sh")
                continue
            if (w["streams"], w["blocks"], w["block_size"]) == (n_streams, n_blocks, cfg.block_size) \
                    and not os.environ.get("RD_K1_IMPL"):
                traffic = int(tj["traffic_bytes"])
                traffic_note = None
            break
    except (OSError, KeyError, ValueError, TypeError, ImportError):
        pass

    traffic_source = None
    if traffic is not None:
        traffic_source = "file: the committed profiles/rNN_traffic.json whose kernel stamp equals the library's"
    if rank == 0 and world == 1 and args.live_traffic and not os.environ.get("RD_K1_IMPL"):
        lt, why = live_traffic(args)
        if lt is not None:
            traffic, traffic_note = lt, None
            traffic_source = ("live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two child runs of this script on this box "
                              "(FETCH_SIZE x 2: gfx950 correction)")
        elif traffic is not None:
            traffic_source += f" (live measurement skipped: {why})"
        else:
            traffic_note = (traffic_note + "; " if traffic_note else "") + f"live measurement skipped: {why}"
    if rank == 0:
        samples_step = n_streams * n_samples
        ms_step = elapsed / args.steps * 1e3
        value = world * samples_step * args.steps / elapsed / 1e6
        dm = float(tm["demod_ms"])
        achieved = samples_step * 2 / (dm * 1e-3) / 1e9
        cnt = bd.counters()
        out = {
            "metric": "complex MSamples/s demodulated (uint8 IQ -> packed bits -> preamble matches -> packets)",
            "value": round(value, 1), "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (FIR: f16 x f16 -> f32 on the matrix pipe, exact integer accumulation)", "data": "synthetic",
            "config": {"workload": f"{n_streams} streams x {n_blocks} blocks x 8192 samples uint8 IQ per GPU "
                                   f"(BASELINE configs[3]/[4]: {UNIQUE} unique synthetic streams tiled, "
                                   f"{in_bytes / 1e9:.2f} GB resident in HBM), 14 samples/symbol, 19.2 kbit/s",
                       "streams_per_gpu": n_streams, "samples_per_stream": n_samples, "parallelism": f"streams/{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "k_demod_bits" if os.environ.get("RD_K1_IMPL", "").startswith("v") else "k_demod_mfma",
                         "kernel_ms": round(dm, 4), "algorithmic_bytes_per_launch": samples_step * 2},
            "kernels_ms": {k[:-3]: (round(float(tm[k]), 4) if tm[k] > 0 else None)   # (pipelined runs: no end-of-run event)
                           for k in (("demod_ms", "fixup_ms", "search_ms", "slice_ms", "total_ms") if args.stage_times
                                     else ("demod_ms", "total_ms"))},
            "completion": "pipelined" if args.pipelined else "per run",
            "tail": "separate kernels + host ordering" if os.environ.get("RD_TAIL_IMPL", "").startswith("l") else "k_tail (one launch)",
            "fixup_runs_frac": round(cnt["fixup_runs"] * 32 / (n_streams * n_samples), 5),
            "packets_per_step": len(recs), "verified_vs_reference_fixtures": verified,
            "h2d_s": round(t_h2d, 3),
        }
        out["settle"] = {"seconds": round(settle_s, 3), "steps": settle_steps,
                         "note": "untimed steps in front of the warmup (clock ramp after the idle upload phase)"}
        if traffic_note:
            out["roofline"]["traffic_note"] = traffic_note
        if traffic_source:
            out["roofline"]["traffic_source"] = traffic_source
        if sustained:
            out["sustained"] = sustained
        if host_fed is not None:
            out["host_fed"] = host_fed
        if per_gpu is not None:
            out["per_gpu"] = per_gpu
            out["dist_backend"] = dist_backend + (f" ({dist_note})" if dist_note else "")
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            # the GPU figures are complete here: show them (stderr) before the CPU legs take their ~25 s; the ONE JSON
            # line on stdout follows with `cpu_baseline` in it
            print("bench.py: GPU part done, CPU baseline legs running: " + json.dumps(out), file=sys.stderr, flush=True)
            out["cpu_baseline"] = cpu_baseline(uniq)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
