import sys, time, torch
sys.path.insert(0, "/root/repo")
from mustafar_amd import compression
from tools.microbench import CFG
dev = torch.device("cuda:0")
for name in ["c3", "c4"]:
    Hq, Hkv, s, Lseq, batch = CFG[name]
    T = ((Lseq - 32) // 256) * 256
    pr = compression.prune_magnitude(torch.randn((batch * Hkv, T, 128), device=dev).half(), s)
    for which in ("key", "value"):
        for sync_copy in (False, True):
            compression._CONVERT_SYNC_COPY = sync_copy
            ts = []
            f0 = compression.convert_fallbacks
            for i in range(120):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                compression._convert(pr, which, onepass=True)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
            ts = ts[10:]
            srt = sorted(ts)
            print(name, which, "copy" if sync_copy else "mirror", "fallbacks", compression.convert_fallbacks - f0, "p10 %.1f p50 %.1f p90 %.1f max %.1f" % (srt[11], srt[55], srt[99], srt[-1]),
                  "slow calls at", [i for i, t in enumerate(ts) if t > 2 * srt[55]][:12], flush=True)
