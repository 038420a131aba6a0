"""Build-time checks on the generated gfx950 ISA (no GPU needed: hipcc cross-compiles to assembly text in a few seconds per file).

Hand-counted `s_waitcnt vmcnt(N)` (csrc/wino_gemm_wide.hip, csrc/conv_bf16d.hip): vmcnt retires in order, so such a wait covers an LDS-DMA piece
(`buffer_load ... lds`) only if AT LEAST N younger loads stand between the piece and the wait -- fewer (a load the compiler merged, sank or
dropped) and the barrier behind the wait would release consumers onto stale LDS bytes, silently.  The check walks every inline-asm vmcnt(N > 0)
of the device code back to the nearest DMA piece or basic-block label and counts the plain loads in between."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "casapose_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def device_asm(source, tmp_path, extra=()):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("no hipcc")
    out = os.path.join(str(tmp_path), os.path.basename(source) + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-munsafe-fp-atomics", "-fno-slp-vectorize",
           "--cuda-device-only", "-S", os.path.join(CSRC, source), "-o", out] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read().splitlines()


def hand_waits(lines):
    """[(line number, N, plain loads between the nearest older DMA piece / block label and the wait, what stopped the walk)]"""
    found = []
    in_asm = False
    for i, ln in enumerate(lines):
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)", t)
        if not (in_asm and m):
            continue
        n = int(m.group(1))
        k, stop = 0, "start"
        j = i
        while j > 0:
            j -= 1
            u = lines[j].strip()
            lab = re.match(r"^([.\w$]+):", u)
            if lab:                                        # a label: other paths join here
                # Round 6: a label reached only by FORWARD branches closes a region that some paths skip -- the loads inside it are not counted (and a
                # DMA piece inside it ends the walk), the walk continues in front of the earliest of those branches, where every path has passed.
                # A label that a LATER instruction branches to is a loop header: the walk ends there, as before.
                name = lab.group(1)
                srcs = [q for q, v in enumerate(lines) if re.match(r"\s*s_cbranch\w*\s+" + re.escape(name) + r"\s*$", v) or re.match(r"\s*s_branch\s+" + re.escape(name) + r"\s*$", v)]
                if not srcs:
                    continue                               # fall-through only
                if max(srcs) > j:
                    stop = "label"
                    break
                first = min(srcs)
                if any((" lds" in lines[q]) and lines[q].strip().startswith(("buffer_load", "global_load")) for q in range(first, j)):
                    stop = "dma"
                    break
                j = first                                  # (the branch itself is no memory operation)
                continue
            if u.startswith(("buffer_load", "global_load")):
                if u.endswith(" lds") or " lds " in u:
                    stop = "dma"
                    break
                k += 1
            if u.startswith(("buffer_store", "global_store", "buffer_atomic", "global_atomic")):
                k += 1                                     # gfx9: stores and atomics count in vmcnt too
        found.append((i + 1, n, k, stop))
    return found


@pytest.mark.parametrize("source,expected_waits", [("wino_gemm_wide.hip", 3), ("conv_bf16d.hip", 2)])
def test_hand_counted_waits_cover_their_dma(tmp_path, source, expected_waits):
    lines = device_asm(source, tmp_path)
    waits = [w for w in hand_waits(lines) if w[1] > 0]
    assert len(waits) >= expected_waits, waits    # (an instantiated template repeats its waits)
    for line, n, younger, stop in waits:
        assert younger >= n, "%s:%d: s_waitcnt vmcnt(%d) with only %d younger VMEM operations since the nearest %s" % (source, line, n, younger, stop)


def test_safe_waits_switch_removes_every_counted_wait(tmp_path):
    lines = device_asm("wino_gemm_wide.hip", tmp_path, extra=["-DCP_SAFE_WAITS"])
    assert all(n == 0 for _, n, _, _ in hand_waits(lines)), hand_waits(lines)


def test_ransac_vote_loop_matches_the_instruction_count_the_bench_line_prices(tmp_path):
    """bench.py --mode vote prices the RANSAC voter against the fp32 VALU issue rate with RANSAC_VALU_PER_HYPOTHESIS vector instructions per hypothesis
    (nine cosine tests): a number read off the generated ISA once.  Re-read it from every build: the loops of vote_kernel<64> that hold
    exactly nine ballot counts (s_bcnt1_i32_b64: one per keypoint test) are the hypothesis loop's common-path copies; their v_* instruction counts must
    bracket the constant."""
    import sys

    sys.path.insert(0, ROOT)
    import bench

    lines = device_asm("ransac_vote.hip", tmp_path)
    start = [i for i, l in enumerate(lines) if re.match(r"^_ZN\S*vote_kernelILi64E\S*:", l)]
    assert start, "vote_kernel<64> not found in the device assembly"
    s = start[0]
    e = next(i for i in range(s, len(lines)) if "s_endpgm" in lines[i])
    body = lines[s:e]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    counts = []
    for i, l in enumerate(body):
        m = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s*s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:      # a backward branch closes a loop
            seg = body[labels[m.group(1)]:i + 1]
            if sum(1 for x in seg if "s_bcnt1_i32_b64" in x) == 9:
                counts.append(sum(1 for x in seg if re.match(r"\s*v_", x)))
    assert counts, "no loop with nine ballot counts: the voting loop changed shape -- recount bench.RANSAC_VALU_PER_HYPOTHESIS"
    assert min(counts) - 3 <= bench.RANSAC_VALU_PER_HYPOTHESIS <= max(counts) + 3, (counts, bench.RANSAC_VALU_PER_HYPOTHESIS)
