# a fresh process per run over a 3 GiB-class file on /dev/shm: where its wall clock goes
set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import blazeseq_amd as B
ctx = B.Context(B.ParserConfig(), "generic", 4096, 0)
n = ctx.generate_synthetic_device(14_700_000, 100, 33, 73, "generic", 0, 0, first=0, count=14_700_000, max_len=100)
buf = torch.empty(n + (1 << 20), dtype=torch.uint8, device="cuda")
ctx.generate_synthetic_device(14_700_000, 100, 33, 73, "generic", buf.data_ptr(), buf.numel(), first=0, count=14_700_000, max_len=100)
host = buf[:n].cpu().numpy()
t0 = time.time(); host.tofile("/dev/shm/bzq_probe.fastq"); print("file", n, "bytes written in", round(time.time() - t0, 2), "s")
PY
for i in 1 2 3 4; do
  S=$EPOCHREALTIME; BZQ_THROUGHPUT_TIMES=1 tests/c_driver/bzq_throughput /dev/shm/bzq_probe.fastq batches 2>&1 | tr '\n' ' '; E=$EPOCHREALTIME; echo " wall $(awk "BEGIN{print ($E - $S) * 1000}") ms"
done
for c in 64 128; do S=$EPOCHREALTIME; BZQ_THROUGHPUT_TIMES=1 tests/c_driver/bzq_throughput /dev/shm/bzq_probe.fastq batches $c 2>&1 | tr '\n' ' '; E=$EPOCHREALTIME; echo " wall $(awk "BEGIN{print ($E - $S) * 1000}") ms"; done
rm -f /dev/shm/bzq_probe.fastq
