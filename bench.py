#!/usr/bin/env python
"""bench.py -- Groth16 prove throughput (constraints/s) of the HIP prover on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2n 20] [--workload prove|msm_g1|msm_sharded]

A "step" is one full groth16.GenerateProofs (groth16/groth16.go:225-278: H(x) = P(x)/Z(x), the
five MSMs, the O(1) tail) over a synthetic instance with n = 2^log2n constraints, m = n + 1
variables, NPublic = 1 (BASELINE.json configs[2]); the proving key, w and px are resident in HBM
before the timed region.  N > 1 (launched by torch.distributed.run, one rank per GPU): every rank
proves its own independent instance of the same size -- the batch-of-proofs partition of
BASELINE.json configs[4]; no data-path collective -- so scaling is weak and `value` is
N * n * K / (max-over-ranks time).  `--workload msm_sharded` instead shards ONE G1 MSM of
N * 2^log2n terms across the ranks with an all-gather of the per-rank partial points
(configs[3]).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline:     the dominant kernel (G1 bucket accumulation) against the HBM roofline,
  cpu_baseline: the reference algorithm (oracle/gs_oracle.c: naive MulScalar/Add loops + schoolbook
                Div) timed on ONE host core on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

import gosnark_amd  # noqa: F401
from gosnark_amd import capi, groth16, synth

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
G1_TERM_BYTES = 96             # 32 B scalar + 64 B affine base per G1 MSM term
G2_TERM_BYTES = 160
R = groth16.R


def checker_leg_proof(proof, inst, r_, s_):
    """Checker leg (with cpu_baseline below the only users of oracle/ in this file; never inside the timed region):
    PiA, PiB, PiC of the benchmarked instance against a*G1, b*G2, c*G1 computed by the C oracle's MulScalar from the
    closed-form scalars of synth.SqchainSetupInstance.expected_proof_scalars."""
    from oracle import c_oracle as C, ref_py as O
    ea, eb, ec = inst.expected_proof_scalars(r_, s_)
    ok = ((proof.PiA[0], proof.PiA[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, ea)) and
          (proof.PiB[0], proof.PiB[1]) == C.g2_affine(C.g2_mul_scalar(O.G2_GEN, eb)) and
          (proof.PiC[0], proof.PiC[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, ec)))
    if not ok:
        raise SystemExit("bench.py: the proof of the benchmarked instance does not match its closed form")
    return "PiA, PiB, PiC equal a*G1, b*G2, c*G1 for the closed-form (a, b, c) derived from the setup's toxic values"


def cpu_baseline_all_cores(log2n_sample, seed):
    """Same algorithm with the term ranges of every MSM split over all host threads (oracle_msm_naive_mt); the reference
    itself is single-threaded (no goroutines), so this is an upper bound on what its algorithm gets from the host."""
    from oracle import c_oracle as C
    threads = os.cpu_count() or 1
    n = 1 << log2n_sample
    inst = synth.random_instance(n, seed)
    g1 = {k: capi.g1_download(inst.g1[k]) for k in ("at", "bacgamma", "bacdelta", "ptd")}
    g2 = capi.g2_download(inst.g2_bacgamma)
    w, hx = inst.w_host, inst.px_host[:n]          # the O(n^2) schoolbook Div is left out here (it does not thread)
    t0 = time.perf_counter()
    C.g1_msm_naive(g1["at"], w, threads=threads)
    C.g1_msm_naive(g1["bacgamma"], w, threads=threads)
    C.g2_msm_naive(g2, w, threads=threads)
    C.g1_msm_naive(g1["bacdelta"][2:], w[2:], threads=threads)
    C.g1_msm_naive(g1["ptd"], hx, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "constraints/s", "cores": threads, "kind": "port",
            "sample": "the five naive MSMs of a Groth16 prove at n=2^%d (no Div), term ranges over %d threads, %.2f s" % (log2n_sample, threads, dt)}


def cpu_baseline(log2n_sample, seed):
    """The reference algorithm on one host core: sum_i MulScalar(base_i, w_i) loops
    (groth16.go:243-250,269-271 / g1.go:140-155 + :32-89) and schoolbook Div (r1csqap.go:70-84),
    via oracle/gs_oracle.c, on an instance with 2^log2n_sample constraints."""
    from oracle import c_oracle as C            # checker/baseline only; never on the product path
    n = 1 << log2n_sample
    m = n + 1
    inst = synth.random_instance(n, seed)       # device arrays -> downloaded Jacobian copies for the CPU
    g1 = {k: capi.g1_download(inst.g1[k]) for k in ("at", "bacgamma", "bacdelta", "ptd")}
    g2 = capi.g2_download(inst.g2_bacgamma)
    w, px, z = inst.w_host, inst.px_host, inst.z_host
    t0 = time.perf_counter()
    hx, _ = C.poly_div_u64(px, z)
    C.g1_msm_naive(g1["at"], w)
    C.g1_msm_naive(g1["bacgamma"], w)
    C.g2_msm_naive(g2, w)
    C.g1_msm_naive(g1["bacdelta"][2:], w[2:])
    C.g1_msm_naive(g1["ptd"][:hx.shape[0]], hx)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "constraints/s", "cores": 1, "kind": "port",
            "sample": "full Groth16 prove at n=2^%d constraints (4 G1 + 1 G2 naive double-and-add MSMs of ~n terms "
                      "+ schoolbook Div), oracle/gs_oracle.c, 1 thread, %.1f s" % (log2n_sample, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--workload", default="prove", choices=["prove", "prove_from_r1cs", "prove_sharded", "prove_pinocchio", "msm_g1", "msm_sharded"],
                    help="prove: one independent proof per GPU (weak scaling, the default the driver runs); prove_sharded: ONE proof "
                         "whose MSM term ranges are split over the ranks (strong scaling, all-gather of 5 partial points); "
                         "prove_from_r1cs: every step also rebuilds px from the resident sparse R1CS and witness (gs_r1cs_px) -- the "
                         "stage upstream of GenerateProofs, reported for information")
    ap.add_argument("--instance", default="setup", choices=["setup", "sqchain", "random"],
                    help="setup: sqchain R1CS + structured trusted setup on the device + px from the sparse system (a complete, "
                         "checkable instance, SURVEY 8d); sqchain: same R1CS with key points k_i*G; random: uniform w / px")
    ap.add_argument("--pipeline", type=int, default=3, choices=[1, 2, 3],
                    help="operations in flight per GPU (prove / msm_g1 workloads): >= 2 = gs_groth16_prove_begin/_end (gs_msm_g1_begin/"
                         "gs_msm_end), the next operation's plan and accumulations are queued behind the current one's; 1 = one "
                         "blocking call per step")
    ap.add_argument("--no-check", action="store_true", help="skip the closed-form proof check of the `setup` instance")
    ap.add_argument("--cpu-log2n", type=int, default=13, help="constraints of the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path for the product code")
    # development aid for 1-GPU boxes: GS_BENCH_SHARE_GPU=1 maps every rank to device 0 and uses gloo, so the multi-process
    # code path (barriers, max-over-ranks timing, the partial-point all-gather) can be exercised without a second GPU
    share = os.environ.get("GS_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    capi.init(local)

    n = 1 << args.log2n
    seed = 0x5EED0002 + (0 if args.workload == "prove_sharded" else rank)
    sharded = args.workload == "prove_sharded"
    from_r1cs = args.workload == "prove_from_r1cs"
    if sharded or from_r1cs:
        args.workload = "prove"
    if args.workload == "prove":
        inst = (synth.sqchain_setup_instance(n, seed) if args.instance == "setup" else
                synth.sqchain_instance(n, seed) if args.instance == "sqchain" else synth.random_instance(n, seed))
        pk = inst.device_pk()
        if sharded and world > 1:
            # each GPU keeps only its 1/world slice of every proving-key array (and builds window tables for that
            # slice only); the full key this rank built for the setup is released
            full = pk
            pk = groth16.ShardPk(full, rank, world)
            full.handle.free()
        r_, s_ = synth.field_elems(2, seed ^ 0xABCDEF, R)

        dev_r1cs = None
        if from_r1cs:
            from gosnark_amd import r1csqap
            dev_r1cs = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)

        def step():
            if sharded:
                return groth16.prove_sharded(pk, inst.w, inst.px, r_, s_)
            if from_r1cs:
                # one call: px from the resident sparse system (overwriting the resident px) behind the accumulations over w
                return groth16.prove_from_r1cs(pk, dev_r1cs, inst.w, r_, s_, inst.px)[0]
            return groth16.prove_resident(pk, inst.w, inst.px, r_, s_)
        units_per_step = n
        workload = ("groth16_prove_2^%d_constraints_sharded_over_all_gpus" if sharded else
                    "groth16_px_from_sparse_r1cs_then_prove_2^%d_constraints_per_gpu" if from_r1cs else
                    "groth16_prove_2^%d_constraints_per_gpu") % args.log2n
    elif args.workload == "prove_pinocchio":
        # snark.GenerateProofs (snark.go:254-289): 6 G1 MSMs over w sharing one plan + 1 G2 MSM + px / Z + 1 G1 MSM over h
        from gosnark_amd import snark
        inst = synth.sqchain_pinocchio_instance(n, seed)
        pk = inst.device_pk()

        def step():
            return snark.prove_resident(pk, inst.w, inst.px)
        units_per_step = n
        workload = "pinocchio_prove_2^%d_constraints_per_gpu" % args.log2n
    else:
        from gosnark_amd import parallel
        nterms = n
        bases = capi.g1_fixed_base(synth.scalars_u64(nterms, seed))
        sc = capi.scalars_upload(synth.scalars_u64(nterms, seed + 77))
        if args.workload == "msm_g1":
            def step():
                return capi.msm_resident(bases, sc, nterms)
        else:
            def step():
                return parallel.msm_g1_sharded(bases, sc, nterms)
        units_per_step = nterms
        workload = ("g1_msm_2^%d_terms_per_gpu" % args.log2n) + ("_allgather_partials" if args.workload == "msm_sharded" else "")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pipelined = args.workload == "prove" and not sharded and not from_r1cs and args.pipeline >= 2
    pin_pipe = args.workload == "prove_pinocchio" and args.pipeline >= 2
    msm_pipe = args.workload == "msm_g1" and args.pipeline >= 2

    def run_steps(count, on_done=None):
        if msm_pipe:
            tickets = []
            for _ in range(count):
                tickets.append(capi.msm_begin(bases, sc, nterms))
                if len(tickets) == args.pipeline:
                    capi.msm_end(tickets.pop(0))
                    if on_done:
                        on_done()
            while tickets:
                capi.msm_end(tickets.pop(0))
                if on_done:
                    on_done()
            return
        if pin_pipe:
            from gosnark_amd import snark as _sn
            tickets = []
            for _ in range(count):
                tickets.append(_sn.prove_begin(pk, inst.w, inst.px))
                if len(tickets) == args.pipeline:
                    _sn.prove_end(tickets.pop(0))
                    if on_done:
                        on_done()
            while tickets:
                _sn.prove_end(tickets.pop(0))
                if on_done:
                    on_done()
            return
        if not pipelined:
            for _ in range(count):
                step()
                if on_done:
                    on_done()
            return
        tickets = []
        for _ in range(count):
            tickets.append(groth16.prove_begin(pk, inst.w, inst.px, r_, s_))
            if len(tickets) == args.pipeline:
                groth16.prove_end(tickets.pop(0))
                if on_done:
                    on_done()
        while tickets:
            groth16.prove_end(tickets.pop(0))
            if on_done:
                on_done()

    run_steps(args.warmup)
    tm_acc = {"acc_g1_ms": 0.0, "acc_g1_launches": 0, "acc_g1_terms": 0, "acc_g2_ms": 0.0, "acc_g2_terms": 0,
              "total_ms": 0.0, "plan_ms": 0.0, "accumulate_ms": 0.0, "reduce_ms": 0.0, "poly_ms": 0.0}
    barrier()
    t0 = time.perf_counter()
    def book():
        tm = capi.last_timing()      # HIP-event timings recorded on the library's streams for the proof just collected
        for k in tm_acc:
            tm_acc[k] += tm[k]
    run_steps(args.steps, book)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    proof_verified = None
    if args.workload == "prove" and args.instance == "setup" and not args.no_check:
        # Product verifier (groth16.VerifyProof -> gs_groth16_verify, host side), outside the timed region, on EVERY rank:
        # the proof of this rank's instance against the vk its device setup produced, for the right public input and a wrong one.
        x_pub = capi.u64_to_ints(inst.w_host[1:2])[0]
        p_last = step()
        good = groth16.VerifyProof(inst.vk, p_last, [x_pub]) and not groth16.VerifyProof(inst.vk, p_last, [(x_pub + 1) % R])
        if world > 1:
            t = torch.tensor([1.0 if good else 0.0], dtype=torch.float64, device="cpu" if share else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            good = bool(t.item() == 1.0)
        if not good:
            raise SystemExit("bench.py: groth16.VerifyProof rejected the proof of the benchmarked instance (or accepted a wrong public input)")
        proof_verified = "groth16.VerifyProof accepted each rank's proof against its device-built vk and rejected a wrong public input (%d/%d ranks)" % (world, world)
    if args.workload == "prove_pinocchio" and not args.no_check:
        from gosnark_amd import snark as _snark
        p_last = step()
        good = _snark.VerifyProof(inst.vk, p_last, inst.public) and not _snark.VerifyProof(inst.vk, p_last, [(inst.public[0] + 1) % R])
        if world > 1:
            t = torch.tensor([1.0 if good else 0.0], dtype=torch.float64, device="cpu" if share else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            good = bool(t.item() == 1.0)
        if not good:
            raise SystemExit("bench.py: snark.VerifyProof rejected the proof of the benchmarked instance (or accepted a wrong public input)")
        proof_verified = "snark.VerifyProof (five pairing equations) accepted each rank's proof against its device-built vk and rejected a wrong public input (%d/%d ranks)" % (world, world)
    proof_check = None
    if rank == 0 and world == 1 and args.cpu_log2n > 0 and args.workload == "prove" and args.instance == "setup" and not args.no_check:
        # Outside the timed region, part of the checker/baseline leg (the only place bench.py touches oracle/): the toxic
        # values of the synthetic setup are known, so the proof the benchmarked instance must produce is known in closed form.
        proof_check = checker_leg_proof(step(), inst, r_, s_)
    host_ms = None
    if rank == 0 and world == 1 and args.workload == "prove":
        # the boundary also accepts HOST buffers (gs_groth16_prove): w (32 B x m) and px (32 B x (2n-1)) then cross PCIe
        # inside the call.  Reported beside the resident-input figure, never as `value`.
        lib = capi.load_library()
        import ctypes
        outp = np.zeros(32, dtype=np.uint64)
        infp = (ctypes.c_int * 3)()
        rs = capi.ints_to_u64([r_, s_])
        th = time.perf_counter()
        for _ in range(3):
            capi.check(lib.gs_groth16_prove(capi.Handle(pk.handle.h), capi.ptr64(inst.w_host), inst.w_host.shape[0], capi.ptr64(inst.px_host),
                                            inst.px_host.shape[0], capi.ptr64(rs[0]), capi.ptr64(rs[1]), capi.ptr64(outp), infp))
        host_ms = (time.perf_counter() - th) / 3 * 1e3

    if rank == 0:
        value = units_per_step * (1 if sharded else world) * args.steps / elapsed
        launches = max(tm_acc["acc_g1_launches"], 1)
        avg_launch_s = tm_acc["acc_g1_ms"] / launches * 1e-3
        bytes_per_launch = G1_TERM_BYTES * tm_acc["acc_g1_terms"] / launches
        achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        is_pin = args.workload == "prove_pinocchio"
        is_prove = args.workload == "prove" or is_pin
        # SURVEY 8d bytes per constraint: Groth16 544 n (4 G1 + 1 G2 MSM) + 128 n (H stage); Pinocchio 7 G1 + 1 G2 + H = 960 n
        step_bytes = (960 * n if is_pin else 672 * n) if is_prove else G1_TERM_BYTES * n
        out = {
            "metric": ("Pinocchio constraints/sec (prove) at 2^%d R1CS" % args.log2n if is_pin else
                       "Groth16 constraints/sec (prove) at 2^%d R1CS" % args.log2n if is_prove else "G1-MSM terms/sec"),
            "value": value,
            "unit": "constraints/s" if is_prove else "terms/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "u32 (9x29-bit Montgomery limbs of the 254-bit BN128 fields)", "data": "synthetic",
            "config": {"workload": workload, "proofs_in_flight": args.pipeline if (pipelined or msm_pipe or pin_pipe) else 1, "constraints": n, "variables": n + 1, "npublic": 1,
                       "parallelism": ("one proof, MSM term ranges sharded over the ranks, all-gather of 5 partial points" if sharded else
                                       "independent proofs, one per GPU") if is_prove else args.workload,
                       "instance": inst.describe() if is_prove else "uniform random scalars, bases k_i*G"},
            "roofline": {"bound": "hbm", "kernel": "k_bucket_accumulate<G1>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "avg_launch_ms": avg_launch_s * 1e3, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "integer-issue bound (254-bit modular arithmetic on 32-bit VALU): PMC evidence in "
                                 "profiles/r01c_pmc_sq_accumulate_g1.txt (VALU ~96 % busy), see DESIGN.md section 5"},
            # the roofline that actually binds the dominant kernel: 32x32+64-bit multiply-add issue (v_mad_u64_u32).  One mixed
            # addition = 6 products (162 mads) + 2 squarings (126) + one two-term product (243) = 1467 mads; a term takes one
            # addition per window.  Peak: 31.5 T lane-mad/s measured by tools/ubench_valu.hip (profiles/r01_ubench_valu.txt).
            "roofline_valu": {"bound": "valu-int-mad", "kernel": "k_bucket_accumulate<G1>",
                              "achieved": (tm_acc["acc_g1_terms"] / launches) * 16 * 1467 / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0,
                              "peak": 31.5, "unit": "T lane-mad/s",
                              "frac": ((tm_acc["acc_g1_terms"] / launches) * 16 * 1467 / avg_launch_s / 1e12 / 31.5) if avg_launch_s > 0 else 0.0,
                              "note": "16 windows at c = 16 (n >= 2^16); other instructions take the remaining issue slots, "
                                      "profiles/r01c_pmc_sq_accumulate_g1.txt"},
            "roofline_whole_step": {"bound": "hbm", "algorithmic_bytes_per_step": step_bytes,
                                    "achieved": step_bytes / (elapsed / args.steps) / 1e9,
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "note": "SURVEY 8d: Groth16 672 B per constraint per proof (544 n MSM + 128 n H stage), Pinocchio 960 B; wall time per step"},
            "device_ms_per_step": {k: tm_acc[k] / args.steps for k in ("total_ms", "poly_ms", "plan_ms", "accumulate_ms", "reduce_ms", "acc_g1_ms", "acc_g2_ms")},
        }
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f).get(workload)
            if pmc:        # measured offline with rocprofv3 --pmc (bench.py cannot attach counters to itself)
                out["roofline"]["traffic"] = pmc["fetch_bytes_per_launch_raw"] + pmc["write_bytes_per_launch_raw"]
                out["roofline"]["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, raw counters)"
        except OSError:
            pass
        if host_ms is not None:
            out["host_buffers_ms_per_step"] = host_ms
        if proof_verified:
            out["proof_verified"] = proof_verified
        if proof_check:
            out["proof_check"] = proof_check
        if world == 1 and args.cpu_log2n > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_log2n, seed + 1000)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.cpu_log2n + 3, seed + 2000)
            # the container may grant fewer CPUs than os.cpu_count() reports: state what the threads actually bought
            out["cpu_baseline_all_cores"]["speedup_vs_1_core"] = out["cpu_baseline_all_cores"]["value"] / out["cpu_baseline"]["value"]
            if args.workload != "prove":
                out["cpu_baseline"]["note"] = "baseline is the Groth16 prove sample; 1 constraint ~ 4 G1 + 1 G2 terms"
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
