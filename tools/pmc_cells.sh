cd $GRAFT_REPO_ROOT; export SVO_SCENE_CACHE=/tmp/svo_scene_cache; mkdir -p $SVO_SCENE_CACHE gpurun_out/r6r
cp profiles/pmc_per_launch.json gpurun_out/pmc_per_launch.json
python tools/pmc_pass.py --tag round6 -- --scene dust --camera K0 > gpurun_out/r6r/pmc_dust_K0.log 2>&1; tail -1 gpurun_out/r6r/pmc_dust_K0.log | cut -c1-120
python tools/pmc_pass.py --tag round6 -- --scene caves --camera CAVE > gpurun_out/r6r/pmc_caves_CAVE.log 2>&1; tail -1 gpurun_out/r6r/pmc_caves_CAVE.log | cut -c1-120
python tools/pmc_pass.py --tag round6 -- --scene terrain --seed 2 --amp 18 --camera K1 > gpurun_out/r6r/pmc_t2a18_K1.log 2>&1; tail -1 gpurun_out/r6r/pmc_t2a18_K1.log | cut -c1-120
python tools/pmc_pass.py --tag round6 -- --camera K2 > gpurun_out/r6r/pmc_t1a8_K2.log 2>&1; tail -1 gpurun_out/r6r/pmc_t1a8_K2.log | cut -c1-120
rm -rf gpurun_out/pmc_round6_*
python - <<PY
import json
j=json.load(open("gpurun_out/pmc_per_launch.json"))
for k,v in j.items():
    o=v.get("other",{})
    print(k, "valu/frame %.3e" % (v["sq_insts_valu"]/4 if "_B4" in k else v["sq_insts_valu"]), "lane util %.3f" % (v["sq_thread_cycles_valu"]/(64*v["sq_active_inst_valu"])), "traffic GB/frame %.3f" % ((v["fetch_size_kb"]*2048+v["write_size_kb"]*1024)/(4 if "_B4" in k else 1)/1e9))
PY
