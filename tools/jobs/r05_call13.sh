# Round 5, thirteenth GPU call: sparse-B split for Pinocchio -- parity (prover module, multi-device module) and the effect at 2^18.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r5m
mkdir -p gpurun_out/$T
(timeout 1500 python -m pytest tests/test_gpu_prove.py tests/test_gpu_zy_multi.py tests/test_gpu_table_policy.py tests/test_gpu_stream_host.py -q --maxfail=5 2>&1 | tail -8) > gpurun_out/$T/pytest.txt; tail -4 gpurun_out/$T/pytest.txt
python - <<'PY' 2>&1 | tee gpurun_out/$T/pinocchio_gates.txt
import time, torch
import gosnark_amd
from gosnark_amd import capi, snark, synth
capi.init(0); capi.set_table_policy("always")
for n in (1 << 18,):
    pin = synth.gates_pinocchio_instance(n, 5)
    pk = pin.device_pk()
    p = snark.prove_resident(pk, pin.w, pin.px)
    assert snark.VerifyProof(pin.vk, p, pin.public)
    for _ in range(2):
        t = [snark.prove_begin(pk, pin.w, pin.px) for _ in range(3)]
        [snark.prove_end(x) for x in t]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tk = []
    for i in range(30):
        tk.append(snark.prove_begin(pk, pin.w, pin.px))
        if len(tk) == 3: snark.prove_end(tk.pop(0))
    while tk: snark.prove_end(tk.pop(0))
    torch.cuda.synchronize()
    print("pinocchio gates 2^18, three in flight: %.3f ms per proof (GS_SPLIT_B_PERCENT=%s)" % ((time.perf_counter() - t0) / 30 * 1e3, __import__('os').environ.get('GS_SPLIT_B_PERCENT', 'default')))
PY
GS_SPLIT_B_PERCENT=0 python - <<'PY' 2>&1 | tee -a gpurun_out/$T/pinocchio_gates.txt
import time, torch
import gosnark_amd
from gosnark_amd import capi, snark, synth
capi.init(0); capi.set_table_policy("always")
n = 1 << 18
pin = synth.gates_pinocchio_instance(n, 5)
pk = pin.device_pk()
for _ in range(2):
    t = [snark.prove_begin(pk, pin.w, pin.px) for _ in range(3)]
    [snark.prove_end(x) for x in t]
torch.cuda.synchronize(); t0 = time.perf_counter()
tk = []
for i in range(30):
    tk.append(snark.prove_begin(pk, pin.w, pin.px))
    if len(tk) == 3: snark.prove_end(tk.pop(0))
while tk: snark.prove_end(tk.pop(0))
torch.cuda.synchronize()
print("pinocchio gates 2^18, three in flight: %.3f ms per proof (GS_SPLIT_B_PERCENT=0: single plan)" % ((time.perf_counter() - t0) / 30 * 1e3))
PY
