#!/bin/bash
# HBM traffic per launch of every benchmarked kernel: for each bench entry, one rocprofv3 --pmc FETCH_SIZE pass and
# one WRITE_SIZE pass over a bench.py run that launches nothing but that entry (the headline run for "headline").
# Output under gpurun_out/$1; tools/traffic_from_pmc.py turns it into profiles/traffic.json.
#   tools/pmc_traffic.sh <tag> [entry ...]
set -e -o pipefail
tag=${1:-traffic}; shift || true
out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
declare -A fam=( [headline]=none [rle_w1]=rle [rle_w8]=rle [rle_w12]=rle [rle_w20]=rle [rle_w12_all_literal]=rle [rle_w12_all_repeated]=rle
  [bss_float32]=bss [bss_float64]=bss [dict_65535_uniform]=dict [dict_65536_zipf_nulls]=dict
  [delta_length_byte_array]=strings [delta_byte_array]=strings [delta_int64_w0]=delta [delta_int64_w64]=delta
  [delta_int64_128x4]=delta [delta_int32]=delta [plain_int64_copy]=delta [delta_encode_int64]=delta )
entries=("$@")
if [ ${#entries[@]} -eq 0 ]; then
  entries=(headline rle_w1 rle_w8 rle_w12 rle_w20 rle_w12_all_literal rle_w12_all_repeated bss_float32 bss_float64
           dict_65535_uniform dict_65536_zipf_nulls delta_length_byte_array delta_byte_array delta_int64_w0
           delta_int64_w64 delta_int64_128x4 delta_int32 plain_int64_copy delta_encode_int64)
fi
for e in "${entries[@]}"; do
  f=${fam[$e]}
  args="--steps 2 --warmup 1 --enc-steps 2 --no-cpu-baseline --no-verify --no-e2e --encodings $f --rle-widths 1,8,12,20"
  if [ "$e" != headline ]; then args="$args --only $e --pages 16"; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== $e $c"
    timeout -s KILL 240 rocprofv3 --pmc $c -d "$out/$e/$c" -o r -- python3 bench.py $args > "$out/$e.$c.log" 2>&1 || { tail -20 "$out/$e.$c.log"; exit 1; }
  done
done
python3 tools/traffic_from_pmc.py "$out" "$out/traffic.json" "$out/traffic.txt"
for e in "${entries[@]}"; do rm -rf "$out/$e"; done      # (raw counter output: the two summaries are what is kept)
