#!/bin/bash
# Same-box A/B of two builds of libvoxhip.so: "$@" is run with the tree's build, then with voxtral_c_amd/libvoxhip_old.so in its place.
cd "$GRAFT_REPO_ROOT" || exit 1
echo "##### new build"; "$@"
cp voxtral_c_amd/libvoxhip.so /tmp/libvoxhip_new.so && cp voxtral_c_amd/libvoxhip_old.so voxtral_c_amd/libvoxhip.so
echo "##### old build"; "$@"
cp /tmp/libvoxhip_new.so voxtral_c_amd/libvoxhip.so
echo "##### new build again"; "$@"
