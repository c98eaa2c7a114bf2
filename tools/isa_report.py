#!/usr/bin/env python3
"""What the compiler made of the chain kernel (csrc/conv_rb16.hip): per variant in use, from the gfx950 ISA --
registers / spilled registers, the vmcnt waits inside the first MFMA loop (is the weight prefetch distance still there?), waits
inside the warm-up touch loops, and the scratch reloads + vmcnt(0) waits between two MFMA loops.  No GPU needed.

  python tools/isa_report.py [-DADK_RB16_PIN_LOADS=0 -DADK_RB16_ASYNC_TOUCH=0 ...]     (extra flags go to hipcc)

profiles/r3_rb16_isa_{head,unpinned_volatile}.txt are its outputs for the final build and for the build before sections 5-6 of
profiles/r3_rb16_timeline.md."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "audiodec_amd", "csrc", "conv_rb16.hip")
HIPCC = "/opt/rocm/bin/hipcc"
# <C, ACT, TA, TB, NTW, SMAX, WPS, PF, EARLY_RES, LS, WR> of the launches of a 256-stream vctk_v1 step (LeakyReLU K11 blocks, ELU K7+1x1 units);
# WR > 0: the round-4 variants with the shared LDS weight ring
IN_USE = {"Li32ELi2ELi11ELi11ELi3ELi1ELi2ELi2ELb0ELi0ELi0E": "vocoder stage 3 (32 ch, 3 tiles, register weight stream)",
          "Li64ELi2ELi11ELi11ELi2ELi1ELi3ELi1ELb1ELi0ELi3E": "vocoder stage 2 (64 ch, weight ring)",
          "Li128ELi2ELi11ELi11ELi2ELi2ELi2ELi3ELb1ELi4ELi0E": "vocoder stage 1 (128 ch)",
          "Li32ELi1ELi7ELi1ELi3ELi1ELi2ELi2ELb1ELi0ELi0E": "encoder block 0 (32 ch, 3 tiles)",
          "Li64ELi1ELi7ELi1ELi2ELi1ELi3ELi1ELb1ELi0ELi3E": "encoder block 1 (64 ch, weight ring)",
          "Li128ELi1ELi7ELi1ELi2ELi2ELi2ELi3ELb1ELi4ELi0E": "encoder block 2 (128 ch)",
          "Li64ELi2ELi11ELi11ELi2ELi1ELi3ELi2ELb1ELi0ELi0E": "(round 3) vocoder stage 2 without the ring: ADK_RB16_RING=0"}


def main():
    flags = sys.argv[1:]
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "rb16.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
               "-Rpass-analysis=kernel-resource-usage", SRC, "-o", asm] + flags
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr[-2000:])
        usage = {}
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = m.group(1); usage[cur] = {}
            for key in ("VGPRs", "VGPRs Spill", "ScratchSize [bytes/lane]"):
                m = re.search(re.escape(key) + r": (\d+)", line)
                if m and cur:
                    usage[cur][key] = int(m.group(1))
        text = open(asm).read()
    parts = re.split(r"\n(_ZN3adk12_GLOBAL__N_116conv_rb16_kernelI[A-Za-z0-9_]+): ; @", text)
    print("hipcc flags:", " ".join(flags) or "(none)")
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1].split("s_endpgm")[0].split("\n")
        tag = next((v for k, v in IN_USE.items() if k in name), None)
        if tag is None:
            continue
        u = usage.get(name, {})
        is_mfma = [l.strip().startswith("v_mfma") for l in body]
        mfma_at = [i for i, f in enumerate(is_mfma) if f]
        # an MFMA loop = a maximal run of MFMAs at most 40 instruction lines apart; what lies between the first two loops is the epilogue
        loops, start = [], mfma_at[0]
        for p_, q_ in zip(mfma_at, mfma_at[1:]):
            if q_ - p_ > 40:
                loops.append((start, p_)); start = q_
        loops.append((start, mfma_at[-1]))
        loops = [lp for lp in loops if sum(is_mfma[lp[0]:lp[1] + 1]) >= 12]
        loop_waits, between, touch_waits = [], [], 0
        for idx, l in enumerate(body):
            t = l.strip()
            m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
            if m and any(lo + (hi - lo) // 4 <= idx <= hi - (hi - lo) // 8 for lo, hi in loops):
                loop_waits.append(int(m.group(1)))            # steady part of a loop: past its first quarter, before its last eighth
            if len(loops) > 1 and loops[0][1] < idx < loops[1][0]:
                if t.startswith("scratch_load"):
                    between.append("reload")
                elif m:
                    between.append(f"vmcnt({m.group(1)})")
                elif t.startswith("s_barrier"):
                    between.append("barrier")
            if "global_load_lds_dword" in t or ("_load_dword " in t and "sc0 sc1" in t):      # LDS-DMA touch / volatile touch
                for t2 in (x.strip() for x in body[idx + 1: idx + 14]):
                    if t2.startswith("s_waitcnt vmcnt"):
                        touch_waits += 1; break
                    if t2.startswith("s_cbranch") or t2.startswith("v_mfma"):
                        break
        lw = loop_waits
        first_loop_end = sum(is_mfma[loops[0][0]:loops[0][1] + 1])
        comp = []
        for e in between:
            if comp and comp[-1][0] == e: comp[-1][1] += 1
            else: comp.append([e, 1])
        early = "residual fetched before" if "Lb1E" in name else "residual fetched after"
        tag = f"{tag}, {early} the second loop"
        print(f"\n{tag}: {u.get('VGPRs', '?')} VGPRs, {u.get('VGPRs Spill', '?')} spilled, {u.get('ScratchSize [bytes/lane]', '?')} B scratch per lane")
        print(f"  vmcnt waits in the steady part of the MFMA loops: min {min(lw) if lw else '-'}, max {max(lw) if lw else '-'}   ({len(loop_waits)} waits; {len(loops)} loops, the first one {first_loop_end} MFMAs)")
        print(f"  warm-up touch sites directly followed by a vmcnt wait: {touch_waits}")
        print("  between the first and the second MFMA loop: " + " ".join(f"{e}x{n}" if n > 1 else e for e, n in comp))


if __name__ == "__main__":
    main()
