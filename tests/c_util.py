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
