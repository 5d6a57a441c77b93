"""Per-basic-block instruction census of a hipcc -S listing: MFMAs, scratch traffic (spills), LDS-DMA, LDS reads, waits, barriers.
Usage: python tools/asm_blocks.py file.s [kernel-substring]"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
kern = None; blocks = []; cur = None
for ln in src:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        kern = m.group(1); cur = None; continue
    if kern is None or want not in kern: continue
    m = re.match(r"^(\.LBB\d+_\d+):", ln)
    if m or cur is None:
        cur = {"k": kern, "name": m.group(1) if m else "entry", "n": 0, "mfma": 0, "scr_ld": 0, "scr_st": 0, "dma": 0, "ds_rd": 0, "ds_wr": 0, "wait": 0, "bar": 0, "gld": 0, "gst": 0, "valu": 0}
        blocks.append(cur)
        if m: continue
    t = ln.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    op = t.split()[0]
    cur["n"] += 1
    if op.startswith("v_mfma"): cur["mfma"] += 1
    elif op.startswith("scratch_load"): cur["scr_ld"] += 1
    elif op.startswith("scratch_store"): cur["scr_st"] += 1
    elif "lds" in t and op.startswith("buffer_load"): cur["dma"] += 1
    elif op.startswith("ds_read") or op.startswith("ds_load"): cur["ds_rd"] += 1
    elif op.startswith("ds_write") or op.startswith("ds_store"): cur["ds_wr"] += 1
    elif op.startswith("s_waitcnt"): cur["wait"] += 1
    elif op.startswith("s_barrier"): cur["bar"] += 1
    elif op.startswith("global_load") or op.startswith("buffer_load"): cur["gld"] += 1
    elif op.startswith("global_store") or op.startswith("buffer_store"): cur["gst"] += 1
    elif op.startswith("v_"): cur["valu"] += 1
    if op == "s_endpgm": kern = None
last = None
for b in blocks:
    if b["k"] != last:
        print("==", b["k"][:110]); last = b["k"]
    if b["n"] >= 20 or b["mfma"]:
        print("  %-12s n=%4d mfma=%3d scr_ld=%3d scr_st=%3d dma=%2d ds_rd=%3d ds_wr=%3d gld=%3d gst=%3d valu=%4d wait=%2d bar=%d" % (
            b["name"], b["n"], b["mfma"], b["scr_ld"], b["scr_st"], b["dma"], b["ds_rd"], b["ds_wr"], b["gld"], b["gst"], b["valu"], b["wait"], b["bar"]))
