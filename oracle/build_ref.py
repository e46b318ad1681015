"""Build the reference's ONLY native component from its own source, where it lies.

  /root/reference/lib/draw_rectangles/draw_rectangles.pyx  ->  oracle/_ref/draw_rectangles*.so

The shipped Cython-0.29 generated C (draw_rectangles.c) does not compile against numpy 2.x,
so the .pyx is re-cythonized with the installed Cython.  Nothing from /root/reference is copied
into the repository: only the compiled shared object stays in oracle/_ref/ (git-ignored; the Cython-generated C -- the
reference's source in another form -- is deleted right after the compile, so what travels to the GPU box with the
snapshot is a binary like this package's own built .so files).  Used to generate tests/golden/raster.npz here; nothing on
the GPU box loads it (the golden file does its job there).
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
PYX = os.environ.get('SGG_REFERENCE', '/root/reference') + '/lib/draw_rectangles/draw_rectangles.pyx'


def build(verbose=True):
    if not os.path.exists(PYX):
        if verbose:
            print('[oracle/_ref] reference tree absent; keeping prebuilt files (if any)')
        return False
    import numpy
    os.makedirs(OUT, exist_ok=True)
    c_file = os.path.join(OUT, 'draw_rectangles.c')
    so = os.path.join(OUT, 'draw_rectangles' + sysconfig.get_config_var('EXT_SUFFIX'))
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(PYX):
        return True
    subprocess.check_call([sys.executable, '-m', 'cython', '-3', PYX, '-o', c_file])
    cmd = ['gcc', '-O2', '-fPIC', '-shared', '-fwrapv', '-Wno-cpp',
           '-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION',
           '-I', sysconfig.get_paths()['include'], '-I', numpy.get_include(), c_file, '-o', so]
    try:
        subprocess.check_call(cmd)
    finally:
        if os.path.exists(c_file):
            os.remove(c_file)            # generated from the reference's source: never kept
    if verbose:
        print('[oracle/_ref] built', so)
    return True


if __name__ == '__main__':
    build()
