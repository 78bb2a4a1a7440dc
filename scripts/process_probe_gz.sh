# fresh-process timing of the three file kinds with the cache's trace on (where a .gz's open goes)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, struct, zlib
sys.path.insert(0, os.getcwd())
import torch
import blazeseq_amd as B
ctx = B.Context(B.ParserConfig(), "generic", 4096, 0)
reads = 14_700_000
n = ctx.generate_synthetic_device(reads, 100, 33, 73, "generic", 0, 0, first=0, count=reads, max_len=100)
buf = torch.empty(n + (1 << 20), dtype=torch.uint8, device="cuda")
ctx.generate_synthetic_device(reads, 100, 33, 73, "generic", buf.data_ptr(), buf.numel(), first=0, count=reads, max_len=100)
host = buf[:n].cpu().numpy()
rb = n // reads
k = (32 << 20) // rb * rb
pb = host[:k].tobytes(); reps = n // k
co = zlib.compressobj(6, zlib.DEFLATED, -15)
member = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3]) + co.compress(pb) + co.flush() + struct.pack("<II", zlib.crc32(pb) & 0xFFFFFFFF, k & 0xFFFFFFFF)
with open("/dev/shm/bzq_probe.fastq.gz", "wb") as f:
    for _ in range(reps): f.write(member)
PY
for i in 1 2 3; do S=$EPOCHREALTIME; BZQ_BUF_CACHE_TRACE=1 BZQ_THROUGHPUT_TIMES=1 tests/c_driver/bzq_throughput /dev/shm/bzq_probe.fastq.gz batches 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; E=$EPOCHREALTIME; echo " wall $(awk "BEGIN{print ($E - $S) * 1000}") ms"; done
rm -f /dev/shm/bzq_probe.fastq.gz
