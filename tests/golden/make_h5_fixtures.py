"""Writes the small HDF5 fixtures of tests/golden/h5/ with h5py -- the library the reference uses -- in the layouts the two
real files have, plus a few variants.  Run once in the build container with the interpreter that has h5py:
    /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py
Every array is also saved to tests/golden/h5/expected.npz, which is what tests/test_hdf5_lite_cpu.py compares the reader with."""
import os
import sys

import h5py
import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'h5')
os.makedirs(HERE, exist_ok=True)
rng = np.random.RandomState(7)
exp = {}

# ---- VG-SGG.h5 (dataloaders/visual_genome.py:536-576): plain contiguous datasets, dtypes as in the released file
n_img, n_box, n_rel = 9, 40, 25
first_box = np.array([0, 5, -1, 9, 14, 20, 27, -1, 33], dtype=np.int32)
last_box = np.array([4, 8, -1, 13, 19, 26, 32, -1, 39], dtype=np.int32)
first_rel = np.array([0, -1, -1, 3, 7, 12, 18, -1, 21], dtype=np.int32)
last_rel = np.array([2, -1, -1, 6, 11, 17, 20, -1, 24], dtype=np.int32)
vg = {
    'split': np.array([0, 0, 0, 2, 0, 2, 0, 0, 2], dtype=np.int32),
    'img_to_first_box': first_box, 'img_to_last_box': last_box, 'img_to_first_rel': first_rel, 'img_to_last_rel': last_rel,
    'labels': rng.randint(1, 151, size=(n_box, 1)).astype(np.int64),
    'boxes_1024': rng.randint(10, 900, size=(n_box, 4)).astype(np.int32),
    'boxes_512': rng.randint(10, 400, size=(n_box, 4)).astype(np.int32),
    'relationships': None,
    'predicates': rng.randint(1, 51, size=(n_rel, 1)).astype(np.int64),
    'active_object_mask': (rng.rand(n_box, 1) > 0.2),
}
rels = np.zeros((n_rel, 2), dtype=np.int32)                 # (subject, object) box ids, inside the image's own box range, s != o
for i in range(n_img):
    if first_rel[i] >= 0:
        for r in range(first_rel[i], last_rel[i] + 1):
            s_, o_ = rng.choice(np.arange(first_box[i], last_box[i] + 1), size=2, replace=False)
            rels[r] = (s_, o_)
vg['relationships'] = rels
with h5py.File(os.path.join(HERE, 'vg_sgg_small.h5'), 'w') as f:
    for k, v in vg.items():
        f.create_dataset(k, data=v)
        exp['vg/' + k] = v.astype(np.uint8) if v.dtype == bool else v

# ---- features.hdf5 (extract_features.py:50-70): one dataset per class, grown row by row, chunk (1, C, P, P), gzip level 4
feat_shape = (6, 3, 3)
with h5py.File(os.path.join(HERE, 'features_small.hdf5'), 'a') as f:
    for name, rows in (('airplane', 5), ('zebra', 1), ('traffic light', 37)):
        for i in range(rows):
            feats = rng.randn(*feat_shape).astype(np.float32)
            if name not in f:
                f.create_dataset(name, data=[feats], maxshape=(None,) + feat_shape, chunks=(1,) + feat_shape, compression=4)
            else:
                d = f[name]
                d.resize(d.shape[0] + 1, axis=0)
                d[-1, :] = feats
        exp['feat/' + name] = f[name][:]

# ---- variants: filters, dtypes, layouts, nesting, many links (multi-level group B-tree), partial edge chunks
with h5py.File(os.path.join(HERE, 'variants.h5'), 'w') as f:
    a = rng.randint(-1000, 1000, size=(37, 11)).astype(np.int16)
    f.create_dataset('shuffle_gzip_i16', data=a, chunks=(8, 4), compression='gzip', shuffle=True)
    exp['var/shuffle_gzip_i16'] = a
    b = rng.rand(5, 7, 3)
    f.create_dataset('chunked_f64_fletcher', data=b, chunks=(2, 7, 2), fletcher32=True)
    exp['var/chunked_f64_fletcher'] = b
    c = rng.rand(12).astype(np.float16)
    f.create_dataset('f16', data=c)
    exp['var/f16'] = c
    d = rng.randint(0, 255, size=(4, 4)).astype(np.uint8)
    f.create_dataset('compact_u8', data=d)                     # small enough for h5py to keep contiguous; still a layout check
    exp['var/compact_u8'] = d
    e = np.arange(10, dtype='>i4')
    f.create_dataset('big_endian_i32', data=e)
    exp['var/big_endian_i32'] = e.astype(np.int32)
    f.create_dataset('scalar', data=np.float32(2.5))
    exp['var/scalar'] = np.float32(2.5)
    f.create_dataset('empty', shape=(0, 4), dtype=np.int32)
    exp['var/empty'] = np.zeros((0, 4), np.int32)
    f.create_dataset('never_written', shape=(6, 5), dtype=np.float32, chunks=(2, 5))
    exp['var/never_written'] = np.zeros((6, 5), np.float32)
    f.create_dataset('names', data=np.array([b'cat', b'zebra', b'dog'], dtype='S5'))
    exp['var/names'] = np.array([b'cat', b'zebra', b'dog'], dtype='S5')
    g = f.create_group('meta').create_group('cls')
    g.create_dataset('ids', data=np.arange(5, dtype=np.int64))
    exp['var/meta/cls/ids'] = np.arange(5, dtype=np.int64)
    many = f.create_group('many')
    for i in range(150):                                        # more links than one symbol-table node holds
        many.create_dataset('d%03d' % i, data=np.array([i, i * i], dtype=np.int32))
    exp['var/many_count'] = np.int64(150)
np.savez(os.path.join(HERE, 'expected.npz'), **exp)
print('wrote', sorted(os.listdir(HERE)), 'with h5py', h5py.__version__, 'HDF5', h5py.version.hdf5_version)
