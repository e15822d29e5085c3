# config 3 per function group (one objective kind per batch), resident route, ms per generation of 16 384 instances:  POP=100 bash tools/exp/lde_groups.sh
for g in "101,102,103,107,108,109" "104,105,106,110,111,112" "113,114,115" "116,117,118" "119,120,121" "122,123,124" "125,126,127" "128,129,130"; do
  echo "$g $(timeout 300 python tools/exp/lde_run.py --route resident --gens-per-launch 50 --steps 100 --pop ${POP:-100} --functions $g 2>&1 | grep -o '"ms_per_generation": [0-9.]*')"
done
