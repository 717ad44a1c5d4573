mkdir -p gpurun_out
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/c4_new.json 2> gpurun_out/c4_new.err
python - <<P
import json
d=json.load(open("gpurun_out/c4_new.json"))
k=d["kernels"]
print("c4 new ms/step", round(d["ms_per_step"],1), "value", round(d["value"]/1e6,2), "align ms/step", round(k["align"]["ms_total"]/k["align"]["steps"],1), "aln/s", round(d["roofline_all"]["align"]["alignments_per_s"]/1e6,1), "nrec", d["records_per_step"])
for n,v in sorted(k.items(), key=lambda kv:-kv[1]["ms_total"]/kv[1]["steps"]): print("  ", n, round(v["ms_total"]/v["steps"],1))
P
