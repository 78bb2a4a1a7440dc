# same-box A/B of two library builds on a fresh PROCESS per run over a .gz of the reference's 100 bp workload:
#   scripts/ab_process_gz.sh ab/old/libblazeseq_hip.so blazeseq_amd/libblazeseq_hip.so
cd $GRAFT_REPO_ROOT
A=$1; B=$2
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
for i in 1 2 3 4 5 6 7 8; do for l in $A $B; do
  S=$EPOCHREALTIME; LD_PRELOAD=$PWD/$l BZQ_THROUGHPUT_TIMES=1 tests/c_driver/bzq_throughput /dev/shm/bzq_probe.fastq.gz batches 2>&1 | grep -v amdgpu.ids | tr '\n' ' ' | cut -c1-220; E=$EPOCHREALTIME
  echo " | $l wall $(awk "BEGIN{print ($E - $S) * 1000}") ms"
done; done
rm -f /dev/shm/bzq_probe.fastq.gz
