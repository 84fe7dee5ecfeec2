#!/bin/bash
# usage: ab_env.sh "ENV=1 ENV2=x" -> the headline bench's key numbers with and without the environment switches, alternating, same box
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
  echo "##### base"; AB_STEPS=${AB_STEPS:-3} bash tools/ab_bench.sh | head -${AB_LINES:-1}
  echo "##### $1"; env $1 AB_STEPS=${AB_STEPS:-3} bash tools/ab_bench.sh | head -${AB_LINES:-1}
done
