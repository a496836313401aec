# The launcher-free multi-device line repeated (N = 2, 4, 8 logical devices, default size) and the N-rank form once more: looking for flakiness, not speed.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4x; mkdir -p $O
for N in 2 4 8 3; do
  S0=$SECONDS; timeout 400 python bench.py --gpus $N --steps 4 --warmup 1 --reps 2 > $O/plain_$N.txt 2>&1; echo "N=$N exit $? wall $((SECONDS-S0)) s"; tail -1 $O/plain_$N.txt
  python - $O/plain_$N.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); s = d.get("strong", {})
        print("   n_gpus %d value %.4g | strong keys %s | watchdog %s | rccl %s" % (d["n_gpus"], d["value"], sorted(s.keys()), s.get("watchdog"), d.get("rccl", {}).get("ranks_seen")))
        for k, v in s.items():
            if isinstance(v, dict) and "error" in v: print("   ERROR in", k, v["error"][:200])
PY
done 2>&1 | tee $O/summary.txt
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 --reps 2 > $O/ranks_2.txt 2>&1; echo "ranks exit $?" | tee -a $O/summary.txt; tail -1 $O/ranks_2.txt | head -c 600 | tee -a $O/summary.txt
