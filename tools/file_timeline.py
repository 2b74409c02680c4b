#!/usr/bin/env python3
"""Kernel timeline of the in-process file path (run on the GPU box):
     cd /tmp && rocprofv3 --kernel-trace -d /tmp/tl -o out --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/file_timeline.py run RECORDS
     python3 tools/file_timeline.py show /tmp/tl
   'show' prints, for the last scan in the trace, the GPU's busy/idle time and the kernels of three chunks in order."""
import csv, glob, os, sys, time

def run(n):
    import ctypes as C
    sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
    from ngs_amd import ffi, host
    lib = ffi.load_library()
    path = b"/tmp/tl.bam"
    if not os.path.exists(path):
        cfg = host.synth_config(n, file_style=int(os.environ.get("STYLE", "0")))   # STYLE=3: an aligner's names, tags and CIGAR mix
        assert lib.ngsq_synth_write_bam(C.byref(cfg), path, n, 6, 0) == 0
        os.sync()
    ctx = host.QcContext([248956422, 242193529], [1, 1], max_read_len=1024, gc_seed=1, sorted_input=True, timing=False, lib=lib)
    for rep in range(3):
        ctx.reset()
        t0 = time.perf_counter()
        h = C.c_void_p()
        assert lib.ngsq_bam_open(path, 0, C.byref(h)) == 0
        got = 0
        while True:
            b = ffi.Batch()
            assert lib.ngsq_bam_next_batch_device(h, ctx._ctx, 1 << 22, C.byref(b)) == 0
            if b.n_records == 0:
                break
            got += int(b.n_records)
            assert lib.ngsq_process_batch(ctx._ctx, C.byref(b), ffi.PASS_BOTH) == 0
        lib.ngsq_bam_close(h)
        ctx.finalize()
        print("scan %d: %d records in %.3f s" % (rep, got, time.perf_counter() - t0), flush=True)

def show(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:28],
                  r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows))
    # the last scan = from the last k_bgzf_inflate-free gap > 50 ms backwards
    infl = [e for e in ev if "bgzf_inflate" in e[2]]
    last = [infl[-1]]
    for e in reversed(infl[:-1]):
        if last[-1][0] - e[1] > 50e6:
            break
        last.append(e)
    t_lo, t_hi = last[-1][0], ev[-1][1]
    scan = [e for e in ev if e[0] >= t_lo]
    busy, cur_s, cur_e = 0, None, None
    for s, e, *_ in scan:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("last scan: %d kernels, %.1f ms from the first inflate to the last kernel, GPU busy %.1f ms (%.0f %%), %d inflate launches"
          % (len(scan), (t_hi - t_lo) / 1e6, busy / 1e6, 100.0 * busy / (t_hi - t_lo), len(last)))
    # cadence: start of every inflate launch and the idle time of the GPU in front of it
    prev_end, line = None, []
    for s_, e_, nm, *_ in scan:
        if "bgzf_inflate" in nm:
            line.append("%.1f(+%.1f)" % ((s_ - t_lo) / 1e6, (s_ - prev_end) / 1e6 if prev_end else 0.0))
        prev_end = max(prev_end or 0, e_)
    print("inflate starts, ms (idle in front):", " ".join(line))
    mid = sorted(last)[len(last) // 2][0]
    end = sorted(last)[min(len(last) - 1, len(last) // 2 + 2)][1]
    print("  start_ms   dur_ms  queue stream kernel")
    for s, e, nm, q, st in scan:
        if mid <= s <= end and e - s > 20000:
            print("  %8.3f %8.3f  %5s %6s %s" % ((s - mid) / 1e6, (e - s) / 1e6, q, st, nm))

if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        show(sys.argv[2])
