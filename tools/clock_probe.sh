# (round 6) what clock does the chip sustain inside the headline step?  rocm-smi samples beside a 200-step bench run.
#   gpurun -- 'bash tools/clock_probe.sh'
cd $GRAFT_REPO_ROOT
python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > gpurun_out/clock_probe_bench.json 2> gpurun_out/clock_probe_bench.err &
BP=$!
sleep 4
for i in $(seq 1 140); do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" | tr '\n' ' ' ; echo
  sleep 0.2
done > gpurun_out/clock_probe_samples.txt
wait $BP
tail -c 600 gpurun_out/clock_probe_bench.json
echo
echo "--- idle"
sleep 3
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo
