#!/bin/bash
# gpurun with the build's identity on board: the snapshot that travels to the GPU box has no .git, so the commit (and whether the
# working tree was clean) is left in .build_commit first -- tools/pmc_summary.py writes it into every counter record, bench.py
# quotes it under roofline.from_profile.commit.
#   tools/gpu.sh --timeout 900 -- '<command>'
cd "$(dirname "$0")/.." || exit 1
if git diff --quiet && git diff --cached --quiet; then state=clean; else state=dirty; fi
echo "$(git rev-parse --short=12 HEAD) $state" > .build_commit
exec /usr/local/graft/bin/gpurun "$@"
