"""Builds the host-side unit-test drivers under tests/host/ with hipcc (no GPU needed: the
__host__ __device__ arithmetic headers are instantiated for the CPU).  Test infrastructure."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
HOST = os.path.join(HERE, "host")
CSRC = os.path.join(os.path.dirname(HERE), "go-snark-study_amd", "csrc")


def build(name, opt="-O2"):
    src = os.path.join(HOST, name + ".hip")
    exe = os.path.join(HOST, name)
    if os.environ.get("GS_HOST_SANITIZE"):      # dev switch: the same drivers under UBSan + ASan (host code only), built aside
        exe += "_san"
        if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-Wno-option-ignored", "-fsanitize=undefined,address",
                                   "-fno-sanitize-recover=undefined", src, "-o", exe])
        return exe
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps):
        return exe
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", opt, "-std=c++17", src, "-o", exe])
    return exe


def run_lines(exe, lines):
    p = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True)
    return p.stdout.split("\n")[:len(lines)]
