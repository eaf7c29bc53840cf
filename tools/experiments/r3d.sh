cd $GRAFT_REPO_ROOT
python tools/experiments/r3d_determinism.py 2>&1 | tail -1
AC_FUSE=0 python tools/experiments/r3d_determinism.py 2>&1 | tail -1
AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_slp.so python tools/experiments/r3d_determinism.py 2>&1 | tail -1
AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_slp.so AC_FUSE=0 python tools/experiments/r3d_determinism.py 2>&1 | tail -1
