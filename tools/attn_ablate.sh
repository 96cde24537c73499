# Timing-only ablation builds of the attention backward (csrc/attention.hip, -DAT_ABL=mask) -> tools/abl/libattn_<mask>.so
# Build here (no GPU needed), run tools/attn_ablate.py on the GPU box.
set -e
cd "$(dirname "$0")/.."
C=pero_pretraining_amd/csrc
for m in 0 1 8 9 32 64 2 4 16; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -Wno-unused-result -Wno-unused-value -DAT_ABL=$m -shared $C/attention.hip $C/api.hip -o tools/abl/libattn_$m.so &
done
wait
ls -la tools/abl
