#!/bin/bash
# builds wakeword_trainer_home_amd/csrc/libwwhip_ab.so = the current objects, except the named sources, which are either taken
# from a git revision (with that revision's headers) or compiled from the working tree with extra flags:
#   tools/build_ab.sh HEAD ww_frontend ww_ctx        tools/build_ab.sh -DWW_LOGMEL_STAMPS ww_frontend
set -e
cd "$(dirname "$0")/../wakeword_trainer_home_amd/csrc"
what=$1; shift
tmp=$(mktemp -d)
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
if [[ $what == -* ]]; then
  for src in "$@"; do hipcc $flags -I../../include -I. $what -c $src.hip -o $tmp/$src.o; done
else
  git -C ../.. archive "$what" wakeword_trainer_home_amd/csrc include | tar -x -C $tmp
  for src in "$@"; do
    (cd $tmp/wakeword_trainer_home_amd/csrc && hipcc $flags -I../../include -I. -c $src.hip -o $tmp/$src.o)
  done
fi
objs=""
for o in *.o; do [[ -f $tmp/$o ]] && objs="$objs $tmp/$o" || objs="$objs $o"; done
hipcc -shared -fPIC --offload-arch=gfx950 -o libwwhip_ab.so $objs
rm -rf $tmp
echo "libwwhip_ab.so: $* from $what"
