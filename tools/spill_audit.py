"""Where do the register spills of the kernels sit?  Compiles every csrc/*.hip to gfx950 assembly (device only, the library's flags)
and reports, per kernel: the resource summary (VGPRs, SGPR spills, scratch bytes) and, for every LOOP that contains matrix
instructions (the K loops), how many SGPR spill moves (v_writelane / v_readlane), scratch accesses and full drains (`s_waitcnt vmcnt(0)`:
those the compiler inserted and those written by hand, told apart by the inline-asm markers) lie INSIDE it.  A scalar spill on a per-tile path costs a few cycles per tile; inside a K loop it would sit in
front of the matrix instructions of every step, and a scratch reload there would drain the LDS-DMA ring.
usage: python tools/spill_audit.py [file.hip ...] > profiles/rNN/spill_audit.txt        (CPU only: hipcc cross-compiles)"""
import os, re, subprocess, sys, tempfile

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "situation_recognition_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-pragma-unroll-threshold=400000", "--cuda-device-only", "-S"]
files = sys.argv[1:] or ["gemm.hip", "c3d.hip", "c3ds.hip", "pair.hip", "expand.hip", "gram.hip", "fp8.hip", "stem.hip"]


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def audit(path):
    with tempfile.TemporaryDirectory() as td:
        s_path = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [path, "-o", s_path], check=True, capture_output=True)
        lines = open(s_path).read().split("\n")
    kernels, cur, name = {}, None, None
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s*; @", ln)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
        elif ln.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            cur.append(ln)
    meta = {}
    for i, ln in enumerate(lines):                     # .amdhsa / remark-style summary comments at each function's end
        m = re.match(r"^; Kernel info:", ln)
    dm = demangle(list(kernels))
    for name, body in kernels.items():
        if not any("v_mfma" in l for l in body):
            continue
        txt = "\n".join(body)
        tail = "\n".join(lines[lines.index(body[-1]) if body else 0:])
        info = {}
        for key in ("NumVgprs", "NumAgprs", "TotalNumSgprs", "ScratchSize", "sgpr_spill_count", "vgpr_spill_count"):
            pass
        # innermost loops: a label whose block is closed by a backward branch to it
        labels = {}
        for i, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = i
        ext = {}                                       # loop header line -> last line that branches back to it
        for i, l in enumerate(body):
            m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"^\s+s_branch\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] <= i:
                ext[labels[m.group(1)]] = max(ext.get(labels[m.group(1)], 0), i)
        mf = [i for i, l in enumerate(body) if "v_mfma" in l]
        best = {}                                      # (first, last matrix instruction inside) -> the tightest loop around them
        for a, b in ext.items():
            inside = [i for i in mf if a <= i <= b]
            if inside:
                k = (inside[0], inside[-1])
                if k not in best or b - a < best[k][1] - best[k][0]:
                    best[k] = (a, b)
        loops = sorted(set(best.values()))
        inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
        rows = []
        # a full drain written by hand sits inside an inline-asm block (;;#ASMSTART .. ;;#ASMEND: the kernels' wait helpers -- e.g. the
        # stream-tail branch of gemm.hip's wait_later -- are asm volatile); one the compiler's wait-count pass inserted does not
        in_asm, asm_flag = False, []
        for l in body:
            if "#ASMSTART" in l:
                in_asm = True
            asm_flag.append(in_asm)
            if "#ASMEND" in l:
                in_asm = False
        for a, b in inner:
            seg = body[a:b + 1]
            nm = sum("v_mfma" in l for l in seg)
            if nm == 0:
                continue
            drains = [i for i in range(a, b + 1) if re.search(r"s_waitcnt vmcnt\(0\)", body[i])]
            rows.append((nm, sum("v_writelane" in l for l in seg), sum("v_readlane" in l for l in seg),
                         sum("scratch_" in l for l in seg), sum(not asm_flag[i] for i in drains), sum(asm_flag[i] for i in drains), b - a + 1))
        tot_w, tot_r, tot_s = txt.count("v_writelane"), txt.count("v_readlane"), txt.count("scratch_")
        print("%s" % dm.get(name, name)[:150])
        print("    whole kernel: %d v_writelane, %d v_readlane (SGPR spill moves), %d scratch instructions" % (tot_w, tot_r, tot_s))
        for nm, w, r, sc, dr, dh, n in rows:
            print("    matrix loop (%4d lines, %3d MFMAs): %d v_writelane, %d v_readlane, %d scratch, `s_waitcnt vmcnt(0)`: %d outside inline asm "
                  "(compiler-inserted), %d inside (hand-placed)" % (n, nm, w, r, sc, dr, dh))


for f in files:
    print("==== %s" % f)
    audit(os.path.join(HERE, f))
