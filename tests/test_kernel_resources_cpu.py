"""Resource budgets the design relies on, checked where the kernels are compiled (hipcc cross-compiles gfx950 without a GPU; `-Rpass-analysis=kernel-resource-usage`):

  * the act kernel that takes late rows (`iqn_qvals_split_kernel<.., LATE = true>`) needs <= 208 registers and no scratch -- two of its wavefronts per SIMD leave 96
    registers, which is what `mn_reset_under_act_kernel` is compiled for;
  * `mn_reset_under_act_kernel`: <= 96 registers, <= 1 408 B of LDS (four of them beside an act workgroup's 154 KB on a 160-KB CU);
  * `mn_reset_kernel` (in front of the act kernel): no scratch;
  * the gradient step `iqn_train_fwdbwd<*, *>`: NO scratch -- the kernel is built to that (csrc/Makefile: values restored from scalar-register spills to scratch were once wrong;
    ADVICE r5), and <= 256 registers.
A compiler or source change that breaks one of these turns into a failing test here instead of a slower (or wrong) run on the GPU."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "distributional_rl_navigation_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I../../include", "-I.", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"]


def _usage(source, flags):
    """{mangled kernel name: {field: int}} from hipcc's resource-usage remarks, compiled with the Makefile's flags for that file."""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    r = subprocess.run([HIPCC] + COMMON + flags + [source], cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


def _pick(usage, *needles):
    ks = [k for k in usage if all(n in k for n in needles)]
    assert ks, (needles, list(usage)[:8])
    return {k: usage[k] for k in ks}


def test_reset_kernels_fit_beside_the_act_kernel():
    u = _usage("mn_reset.hip", ["-ffp-contract=off"])
    for k, v in _pick(u, "mn_reset_under_act_kernel").items():
        assert v["VGPRs"] + v.get("AGPRs", 0) <= 96, (k, v)
        assert v["LDS Size"] <= 1408, (k, v)
    for k, v in _pick(u, "15mn_reset_kernel").items():
        assert v["ScratchSize"] == 0, (k, v)


def test_act_kernel_leaves_96_registers_per_simd():
    u = _usage("iqn_act.hip", ["-ffp-contract=fast", "-fno-slp-vectorize"])
    late = _pick(u, "iqn_qvals_split_kernel", "Lb0ELb0ELi8ELb1E")      # <QUANT = false, SHARED = false, 8 waves, LATE = true>
    plain = _pick(u, "iqn_qvals_split_kernel", "Lb0ELb0ELi8ELb0E")
    for k, v in {**late, **plain}.items():
        assert v["VGPRs"] + v.get("AGPRs", 0) <= 208 and v["ScratchSize"] == 0, (k, v)


def test_gradient_step_has_no_scratch():
    u = _usage("iqn_train.hip", ["-ffp-contract=off", "-mllvm", "-disable-machine-licm"])
    ks = _pick(u, "iqn_train_fwdbwd")
    assert len(ks) == 3      # <XCHG, FUSED> = <false, false>, <false, true>, <true, true>
    for k, v in ks.items():
        assert v["ScratchSize"] == 0 and v["VGPRs"] + v.get("AGPRs", 0) <= 256, (k, v)
