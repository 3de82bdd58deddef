#!/bin/bash
# tools/ab/build_commit.sh <name> <commit>: the whole libe2k.so of another commit -> tools/ab/lib/libe2k_<name>.so (same-box A/B of changes
# that span several source files; build_variants.py swaps ONE file).  E2K_LIB=<path> makes the package load it -- with TODAY's header:
# only for commits whose C ABI is a subset-compatible of today's calls.
set -e
cd "$(dirname "$0")/../.."
name=$1; commit=$2
wt=$(mktemp -d /tmp/e2k_wt_XXXX)
git worktree add -f --detach "$wt" "$commit" > /dev/null 2>&1
python "$wt/e2-tts-pytorch_amd/build_kernels.py" > /dev/null 2>&1
mkdir -p tools/ab/lib
cp "$wt/e2-tts-pytorch_amd/e2_tts_pytorch_amd/libe2k.so" "tools/ab/lib/libe2k_$name.so"
git worktree remove --force "$wt"
echo "tools/ab/lib/libe2k_$name.so"
