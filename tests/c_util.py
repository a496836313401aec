"""Plain-C drivers of the C ABI (tests/c/*.c): what a cgo caller does, from a process without Python or torch.
Every Go wrapper in go/ has its exact call sequence as one of these programs (INTEGRATION.md lists the pairs)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from gosnark_amd import capi
import golden_util as GU
from oracle import ref_py as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CDIR = os.path.join(ROOT, "tests", "c")
LIBDIR = os.path.join(ROOT, "go-snark-study_amd")


def compile_c(name, outdir):
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    exe = os.path.join(str(outdir), os.path.splitext(name)[0])
    subprocess.check_call([gcc, "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", CDIR,
                           os.path.join(CDIR, name), "-o", exe, "-L", LIBDIR, "-lgosnark_hip", "-Wl,-rpath," + LIBDIR, "-lpthread"])
    return exe


def build_and_run(name, args, outdir, timeout=600):
    exe = compile_c(name, outdir)
    run = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=timeout)
    assert run.returncode == 0, "%s failed (%d):\n%s\n%s" % (name, run.returncode, run.stdout, run.stderr)
    return run.stdout


def _blob(path, parts):
    with open(path, "wb") as f:
        for p in parts:
            f.write(np.ascontiguousarray(p, dtype="<u8").tobytes())
    return path


def write_groth_instance(outdir, rec, public=(35,)):
    """The flat-limb instance file the Groth16 drivers read (layout: tests/c/instance.h, read_groth_instance)."""
    from gosnark_amd import utils
    opk = GU.groth_pk(rec["setup"])
    _, vk = utils.GrothSetupFromString(rec["setup"])
    r, s = GU.rs_from_stream(rec["rand"])
    m, npx = len(rec["w"]), len(rec["px"])
    parts = [np.array([m, npx, len(opk.Z), len(opk.PowersTauDelta), len(vk.IC), rec["circuit"]["NPublic"]], dtype=np.uint64),
             capi.g1_points_to_u64(opk.G1_At), capi.g1_points_to_u64(opk.G1_BACGamma), capi.g2_points_to_u64(opk.G2_BACGamma),
             capi.g1_points_to_u64(opk.BACDelta), capi.g1_points_to_u64(opk.PowersTauDelta),
             capi.g1_points_to_u64([opk.G1_Alpha, opk.G1_Beta, opk.G1_Delta]), capi.g2_points_to_u64([opk.G2_Beta, opk.G2_Delta]),
             capi.ints_to_u64([z % O.R for z in opk.Z]), capi.ints_to_u64([x % O.R for x in rec["w"]]),
             capi.ints_to_u64([x % O.R for x in rec["px"]]), capi.ints_to_u64([r, s]),
             capi.g1_points_to_u64([vk.G1_Alpha]), capi.g2_points_to_u64([vk.G2_Beta, vk.G2_Gamma, vk.G2_Delta]), capi.g1_points_to_u64(vk.IC),
             capi.ints_to_u64(list(public))]
    return _blob(os.path.join(str(outdir), "groth_instance.bin"), parts)


def write_pinocchio_instance(outdir, rec, public=(35,)):
    """snark.Pk / Vk / w / px of a Pinocchio golden (layout: tests/c/instance.h, read_pinocchio_instance)."""
    from gosnark_amd import utils
    pk, vk = utils.SetupFromString(rec["setup"])
    m, npx = len(rec["w"]), len(rec["px"])
    g1 = capi.g1_points_to_u64
    parts = [np.array([m, npx, len(pk.Z), len(pk.G1T), len(vk.IC), rec["circuit"]["NPublic"]], dtype=np.uint64),
             g1(pk.A), g1(pk.Ap), capi.g2_points_to_u64(pk.B), g1(pk.Bp), g1(pk.C), g1(pk.Cp), g1(pk.Kp), g1(pk.G1T),
             capi.ints_to_u64([z % O.R for z in pk.Z]), capi.ints_to_u64([x % O.R for x in rec["w"]]), capi.ints_to_u64([x % O.R for x in rec["px"]]),
             capi.g2_points_to_u64([vk.Vka]), g1([vk.Vkb]), capi.g2_points_to_u64([vk.Vkc]), g1([vk.G1Kbg]),
             capi.g2_points_to_u64([vk.G2Kbg, vk.G2Kg, vk.Vkz]), g1(vk.IC), capi.ints_to_u64(list(public))]
    return _blob(os.path.join(str(outdir), "pinocchio_instance.bin"), parts)


def write_r1cs(outdir, dense_abc, npublic, toxic, name="r1cs.bin"):
    """Three dense R1CS matrices ([constraint][variable] ints) as CSR + the toxic values (read_r1cs_instance)."""
    from gosnark_amd import r1csqap
    n, m = len(dense_abc[0]), len(dense_abc[0][0])
    parts = [np.array([n, m, npublic, len(toxic)], dtype=np.uint64)]
    for mat in dense_abc:
        rp, cl, vl = r1csqap.csr_from_rows([{k: v for k, v in enumerate(row) if v} for row in mat])
        parts += [np.array([len(cl)], dtype=np.uint64), rp.astype(np.uint64), cl.astype(np.uint64), vl]
    parts.append(capi.ints_to_u64([t % O.R for t in toxic]))
    return _blob(os.path.join(str(outdir), name), parts)


def read_words(path):
    return np.fromfile(path, dtype="<u8")
