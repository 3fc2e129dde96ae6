import sys, os, glob, subprocess
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
if len(sys.argv) > 2 and sys.argv[1] == "child":
    sys.path.insert(0, root)
    import jpeglibrary_amd as jl
    files = sys.argv[2:]
    for f in files:
        print("FILE", f, flush=True)
        d = open(f, "rb").read()
        try:
            jl.decode_batch([d], jl.FMT_INTERLEAVED_U8)
        except jl.JpegError:
            pass
        print("DEC", flush=True)
        try:
            b = jl.OptimizeBatch().upload([d], True).run()
            b.result(0)
            b.close()
        except jl.JpegError:
            pass
    print("DONE", flush=True)
    sys.exit(0)
files = sorted(glob.glob(os.path.join(here, sys.argv[1], "*.jpg")))
i = 0
while i < len(files):
    chunk = files[i:i + 200]
    p = subprocess.run([sys.executable, __file__, "child"] + chunk, capture_output=True, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("FILE") or l in ("DEC", "DONE")]
    if lines and lines[-1] == "DONE":
        i += len(chunk)
        continue
    last = [l for l in lines if l.startswith("FILE")][-1].split(" ", 1)[1]
    stage = "optimize" if lines[-1] == "DEC" else "decode"
    print("CRASH rc", p.returncode, stage, last, flush=True)
    os.makedirs(os.path.join(root, "gpurun_out", "crash"), exist_ok=True)
    open(os.path.join(root, "gpurun_out", "crash", os.path.basename(last)), "wb").write(open(last, "rb").read())
    i = files.index(last) + 1
print("scan finished")
