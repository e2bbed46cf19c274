set -x
mkdir -p gpurun_out
python tools/phase_profile.py 2>&1 | tail -40
python tools/phase_profile.py --waves-per-member 2 2>&1 | tail -40
bash tools/prof_pmc.sh gpurun_out/pmc1 2>&1 | tail -150
head -3 $(find gpurun_out/pmc1/pass1 -name "*counter_collection.csv" | head -1)
