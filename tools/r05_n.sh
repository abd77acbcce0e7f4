#!/bin/bash
# round 5, session n: flush timeline, then the profile passes on the sources as they stand (re-stamps profiles/traffic.json)
cd "$(dirname "$0")/.." && . tools/env.sh
bash tools/r05_m.sh || exit 1
bash tools/r05_final_b.sh c2 c3 c4 c5
