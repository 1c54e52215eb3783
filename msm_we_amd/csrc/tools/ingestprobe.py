"""Where `stage_iterations` spends its time by block size and source (round 4): the bench's C3-shaped iterations
staged from memory and from the run-level raw files with STAGE_CHUNK_BYTES = 16 / 32 / 64 / 128 MB; prints seconds, GB/s and
the host thread's waits (page-locking, reader fill, uploads, release).  Usage: python ingestprobe.py [n_iterations]"""
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from msm_we_amd import _abi, synthetic  # noqa: E402
from msm_we_amd.data import DataMixin, InMemorySource, NpyDirectorySource  # noqa: E402
from msm_we_amd.model import modelWE  # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 400
h = _abi.Hamsm()
cfg = synthetic.WEConfig("c3", 3, n_it, 10000, 30, 4, 250)
lat = synthetic.latent_centres(cfg)
its = {it: synthetic.make_iteration(cfg, it, lat) for it in range(1, n_it + 2)}
tmp = tempfile.mkdtemp(prefix="hamsm_ingestprobe_")
try:
    raw = NpyDirectorySource.export(InMemorySource(its), os.path.join(tmp, "npy"))
    for rep in range(2):
        for mb, piece_mb, jpr in ((32, 6, 1), (32, 8, 1), (32, 12, 1), (32, 16, 1), (32, 24, 1), (16, 16, 1), (24, 12, 1), (48, 12, 1)):
            DataMixin.STAGE_CHUNK_BYTES = mb << 20
            DataMixin.STAGE_READERS = piece_mb   # (second column: reader threads)
            for name, source in (("npy_files", raw), ("in_memory", InMemorySource(its))):
                m = modelWE(h=h)
                m.pcoord_ndim = 1
                m.set_iteration_source(source)
                m.get_iterations()
                t0 = time.perf_counter()
                st = m.stage_iterations()
                h.sync()
                dt = time.perf_counter() - t0
                print(mb, "MB", piece_mb, jpr, name, json.dumps({"seconds": round(dt, 4), "GBps": round(m.ingest_stats["bytes"] / dt / 1e9, 2),
                                                  "blocks": m.ingest_stats["blocks"], "waits": m.ingest_stats["host_waits_s"]}), flush=True)
                for b in (st.Xp, st.Xc, st.p0, st.p1, st.w):
                    b.free()
finally:
    shutil.rmtree(tmp, ignore_errors=True)
    h.close()
