"""Static check of a hand-scheduled kernel's ISA (hipcc -S output): for every kernel whose name contains the pattern, list what sits inside its
innermost loops -- scratch spills / reloads (a reload's vmcnt wait drains the asm LDS-DMA), hipcc-inserted vmcnt waits, barriers, MFMA
and VALU counts.   python tools/isa_loop_check.py /tmp/attention2.s attn2_kernel"""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pat in l]
    for st in starts:
        end = next(i for i in range(st, len(lines)) if lines[i].strip().startswith("s_endpgm"))
        name = lines[st].split(":")[0]
        # innermost loop = blocks tagged "Depth=2" (Parent Loop / in Loop ... Depth=2)
        inner = [i for i in range(st, end) if "Depth=2" in lines[i]]
        if not inner:
            print(name, "no depth-2 loop")
            continue
        lo, hi = inner[0], None
        # the loop body extends to the last block label mentioning Depth=2, up to the next label after it
        last = inner[-1]
        hi = next((i for i in range(last + 1, end) if re.match(r"^\.LBB", lines[i])), end)
        body = lines[lo:hi]
        cnt = lambda rx: sum(1 for l in body if re.search(rx, l))  # noqa: E731
        waits = [l.strip() for l in body if "s_waitcnt" in l and "vmcnt" in l]
        print(f"{name}: inner loop {hi - lo} lines | mfma {cnt(r'v_mfma')} exp {cnt(r'v_exp_f32')} ds_read {cnt(r'ds_read')} glds {cnt(r'global_load_lds')} "
              f"barrier {cnt(r's_barrier')} | scratch {cnt(r'scratch_')} global_load {cnt(r'global_load_dword')} buffer {cnt(r'buffer_')} | vmcnt waits: {sorted(set(waits))}")


if __name__ == "__main__":
    main()
