#!/bin/bash
# rocprofv3 kernel stats of one-query blocking calls on the embedding shape (200k x 1536, unit-norm, m = 32): the engine's choice (variant 18 for
# M1 / M2, DR_MODE_PQB + rerank). Output: gpurun_out/prof5c/ -> profiles/r05/kernel_stats_latency_embeddings.csv
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof5c; rm -rf $OUT; mkdir -p $OUT
LAT_NQ=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 scripts/latency_embeddings.py > $OUT/latency_embeddings_under_rocprof.json 2> $OUT/stats.err
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_latency_embeddings.csv; rm -rf $OUT/stats
grep -E "lat_kernel|lut_build|rerank_kernel|pqb_kernel|pq_bound|permute_queries|search_kernel<1536" $OUT/kernel_stats_latency_embeddings.csv | cut -c1-170
