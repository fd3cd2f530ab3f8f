#!/bin/bash
# round 5, GPU call 7: ns_generate_step (ABI 6) as the default schedule of bench.py and the CLI — the whole -m gpu suite, the default line,
# the A/B against the Python-thread schedule, the CLI's per-batch device time with the output dropped behind PCIe, k_cs_hist timed.
cd "$(dirname "$0")/../.."
O=gpurun_out/r05h; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) | tee $O/pytest_gpu.log
timeout 400 python bench.py --steps 10 --warmup 3 2>$O/bench_default.err | tail -1 > $O/bench_ecoli_fasta.json
python - $O/bench_ecoli_fasta.json <<'P' | tee $O/bench_default_summary.log
import json,sys
d=json.load(open(sys.argv[1])); r=lambda x:round(x,3)
print("step", r(d["ms_per_step"]), "ms", r(d["value"]/1e6), "M reads/s", d["config"].get("step_call"), "| aligned", r(d["aligned_batch"]["device_ms"]), {k:r(v) for k,v in d["kernel_ms"].items() if v>0.01}, "| unaligned", r(d["unaligned_batch"]["device_ms"]), {k:r(v) for k,v in d["unaligned_batch"]["kernel_ms"].items() if v>0.01})
print("roofline", {k:(r(v) if isinstance(v,float) else v) for k,v in d["roofline"].items() if k in ("frac","frac_kernel_only_bytes","frac_counter_bytes","whole_aligned_batch_frac","traffic_source")})
for k in ("serial","errlog_on"): print(k, {a:(r(b) if isinstance(b,float) else b) for a,b in d[k].items() if not isinstance(b,(dict,str))})
c=d.get("configs2",{}); print("configs2", r(c.get("ms_per_step",0)), c.get("aligned_batch",{}).get("kernel_ms"), c.get("roofline",{}).get("frac"))
print("cpu", d.get("cpu_baseline",{}).get("value"), "e2e", {k:(r(v.get("reads_per_s",0)/1e6) if isinstance(v,dict) and "reads_per_s" in v else None) for k,v in d.get("e2e",{}).items()})
P
for mode in "" "--python-threads" "" "--python-threads"; do echo -n "bench ${mode:-ns_generate_step} "; timeout 150 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-e2e --no-configs2 --no-extras $mode 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=lambda x:round(x,2)
print(r(d['ms_per_step']), 'ms/step', r(d['value']/1e6), 'M reads/s; aligned', r(d['aligned_batch']['device_ms']), '; unaligned', r(d['unaligned_batch']['device_ms']))"; done 2>&1 | tee $O/ab_step_call.log
NS_CLI_DROP_OUTPUT=1 NS_CLI_TRACE=1 timeout 300 python scripts/bench_cli.py -n 8000000 > $O/bench_cli_drop_8M.log 2>&1; tail -22 $O/bench_cli_drop_8M.log
for e in "" "NS_CS_NO_SORT=1"; do echo -n "k_cs_hist ${e:-sorted} "; env $e timeout 200 python scripts/bench_characterize.py --alignments 1000000 2>/dev/null | tail -1; done | tee $O/bench_characterize.log
