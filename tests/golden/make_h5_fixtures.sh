#!/bin/sh
# Regenerates tests/golden/mvp_tiny_*.h5 with the real HDF5 library of the build image.
set -e
here=$(cd "$(dirname "$0")" && pwd)
gcc -O1 -I/opt/conda/include "$here/make_h5_fixtures.c" -L/opt/conda/lib -Wl,-rpath,/opt/conda/lib -lhdf5 -o /tmp/make_h5_fixtures
/tmp/make_h5_fixtures "$here"
ls -l "$here"/mvp_tiny_*.h5
