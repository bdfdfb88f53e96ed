for i in 1 2 3; do
timeout -k 10 120 python scripts/small_p_latency.py > gpurun_out/small_p.json 2>/dev/null
python - <<PY
import json
sp=json.load(open("gpurun_out/small_p.json"))
print({k:round(v["wall_us_per_call"],1) for k,v in sp.items() if "P=1000" in k})
PY
done
