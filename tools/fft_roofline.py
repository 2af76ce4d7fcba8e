"""BASELINE.json configs[2] kernel run: C random M31 columns of N = 2^log cells, time circle iFFT (interpolate) and LDE FFT (evaluate
N -> 2N) through the C ABI; per-kernel GB/s from the library's HIP-event profile. Usage: python tools/fft_roofline.py [log] [C ...]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package, splitmix_column
import numpy as np

def main():
    log = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    cs = [int(x) for x in sys.argv[2:]] or [1, 16, 128]
    pkg = load_package(); lib = pkg.lib()
    ctx = pkg.Context(0, max_log_domain=log + 1)
    out = []
    for C in cs:
        base = splitmix_column(0x5EED0000, 1 << log)
        src = [ctx.upload(np.roll(base, c)) for c in range(C)]
        lde = [ctx.malloc(4 << (log + 1)) for _ in range(C)]
        for rep in range(3):
            if rep == 1:
                lib.bfhip_profile_enable(ctx._h, 1); lib.bfhip_profile_reset(ctx._h)
            ctx.interpolate(src, src, log)
            ctx.evaluate(src, lde, log, log + 1)
            ctx.evaluate(src, src, log, log)      # back to evaluations so values stay random-looking
        js = ctypes.c_void_p(); lib.bfhip_profile_report(ctx._h, ctypes.byref(js))
        rep = json.loads(ctypes.string_at(js).decode()); lib.bfhip_free_host(js); lib.bfhip_profile_enable(ctx._h, 0)
        row = {"log": log, "columns": C}
        for k, v in rep.items():
            row[k] = {"launches": v["calls"], "avg_us": round(v["total_ms"] / v["calls"] * 1e3, 1), "GB/s_moved": round(v["bytes"] / v["total_ms"] / 1e6, 1)}
        tot = sum(v["total_ms"] for v in rep.values()) / 2   # two profiled repetitions
        cells = C * (1 << log)
        row["ifft_plus_lde_plus_fft_ms"] = round(tot, 3)
        # algorithmic bytes of one repetition: iFFT 8N + LDE 12N + same-size FFT 8N per column
        row["algorithmic_GB/s"] = round(28.0 * cells / (tot * 1e-3) / 1e9, 1)
        out.append(row)
        for p in src + lde: ctx.free(p)
    print(json.dumps(out, indent=1))
    ctx.close()

if __name__ == "__main__":
    main()
