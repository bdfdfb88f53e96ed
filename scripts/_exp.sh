timeout -k 10 300 python -m pytest tests/test_10_raster_gpu.py -x -q -k "tile_order or forward_parity or tiny" 2>&1 | tail -3
timeout -k 10 120 python scripts/small_p_latency.py > gpurun_out/small_p.json 2>/dev/null
python - <<PY
import json
sp=json.load(open("gpurun_out/small_p.json"))
for k,v in sp.items(): print(k, round(v["wall_us_per_call"],1), round(v["gpu_us_per_call"],1))
PY
for v in 0 1; do OGS_TILE_ORDER=$v timeout -k 10 120 python scripts/tile_order_ab.py --skew 2.0 2>/dev/null; done
