"""sgg_amd.hdf5_lite against files written by h5py (tests/golden/make_h5_fixtures.py, run with the build container's conda
interpreter): the layout of VG-SGG.h5, the layout of features.hdf5 (row-chunked, gzip, grown by resize) and format variants."""
import os

import numpy as np
import pytest

from sgg_amd import hdf5_lite as H

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'h5')
EXP = dict(np.load(os.path.join(HERE, 'expected.npz')))


def test_vg_sgg_layout_every_dataset_and_the_reference_index_expressions():
    with H.File(os.path.join(HERE, 'vg_sgg_small.h5')) as f:
        names = sorted(k[3:] for k in EXP if k.startswith('vg/'))
        assert sorted(f.keys()) == names
        for k in names:
            d, e = f[k], EXP['vg/' + k]
            assert d.shape == e.shape and len(d) == e.shape[0]
            got = d[:]
            np.testing.assert_array_equal(got.astype(e.dtype), e)
            if k != 'active_object_mask':
                assert got.dtype == e.dtype, (k, got.dtype, e.dtype)
        # dataloaders/visual_genome.py:537-576
        split_mask = f['split'][:] == 2
        split_mask &= f['img_to_first_box'][:] >= 0
        np.testing.assert_array_equal(f['img_to_first_box'][split_mask], EXP['vg/img_to_first_box'][split_mask])
        np.testing.assert_array_equal(f['labels'][:, 0], EXP['vg/labels'][:, 0])
        np.testing.assert_array_equal(f['predicates'][:, 0], EXP['vg/predicates'][:, 0])
        np.testing.assert_array_equal(f['boxes_1024'][3:9], EXP['vg/boxes_1024'][3:9])
        assert 'boxes_512' in f and 'boxes_2048' not in f
        with pytest.raises(KeyError):
            f['nope']


def test_load_graphs_reads_an_h5_path_without_h5py():
    """The VG loader (sgg_amd/visual_genome.py) opens a .h5 path through hdf5_lite and gives what it gives for the same arrays."""
    from sgg_amd import visual_genome as V
    path = os.path.join(HERE, 'vg_sgg_small.h5')
    arrays = {k[3:]: v for k, v in EXP.items() if k.startswith('vg/')}
    for mode in ('train', 'test'):
        a = V.load_graphs(path, mode=mode, num_im=-1, num_val_im=1, filter_empty_rels=True, min_graph_size=-1, max_graph_size=-1,
                          filter_non_overlap=False)
        b = V.load_graphs(arrays, mode=mode, num_im=-1, num_val_im=1, filter_empty_rels=True, min_graph_size=-1, max_graph_size=-1,
                          filter_non_overlap=False)
        np.testing.assert_array_equal(a[0], b[0])
        for xa, xb in zip(a[1:], b[1:]):
            assert len(xa) == len(xb)
            for u, v in zip(xa, xb):
                np.testing.assert_array_equal(u, v)


def test_features_layout_rows_chunks_and_partial_reads():
    with H.File(os.path.join(HERE, 'features_small.hdf5')) as f:
        assert sorted(f.keys()) == sorted(k[5:] for k in EXP if k.startswith('feat/'))
        for k in f.keys():
            d, e = f[k], EXP['feat/' + k]
            assert d.shape == e.shape and d.dtype == np.float32 and d.chunks == (1,) + e.shape[1:]
            np.testing.assert_array_equal(d[:], e)
            np.testing.assert_array_equal(d[-1], e[-1])                      # augment/gan.py samples rows by index
            np.testing.assert_array_equal(d[0, 2], e[0, 2])
            if e.shape[0] > 4:
                np.testing.assert_array_equal(d[[4, 0, 3]], e[[4, 0, 3]])
                np.testing.assert_array_equal(d[1:4], e[1:4])
                np.testing.assert_array_equal(d[np.arange(len(e)) % 2 == 0], e[::2])
            with pytest.raises(IndexError):
                d[len(e)]
        # only the chunks that are asked for are inflated
        d = f['traffic light']
        calls = []
        orig = d._read_chunk
        d._read_chunk = lambda off: (calls.append(off), orig(off))[1]
        d[[7, 30]]
        assert sorted(calls) == [(7, 0, 0, 0), (30, 0, 0, 0)]


def test_format_variants():
    with H.File(os.path.join(HERE, 'variants.h5')) as f:
        for k in ('shuffle_gzip_i16', 'chunked_f64_fletcher', 'f16', 'compact_u8', 'big_endian_i32', 'empty', 'never_written', 'names'):
            e = EXP['var/' + k]
            got = f[k][:] if f[k].shape else f[k][()]
            assert got.shape == e.shape, k
            np.testing.assert_array_equal(got.astype(e.dtype), e, err_msg=k)
        assert float(f['scalar'][()]) == 2.5
        np.testing.assert_array_equal(f['meta/cls/ids'][:], EXP['var/meta/cls/ids'])
        np.testing.assert_array_equal(f['meta']['cls']['ids'][2:], EXP['var/meta/cls/ids'][2:])
        many = f['many']
        assert len(many.keys()) == int(EXP['var/many_count'])
        for i in (0, 77, 149):
            np.testing.assert_array_equal(many['d%03d' % i][:], np.array([i, i * i], dtype=np.int32))
        np.testing.assert_array_equal(f['shuffle_gzip_i16'][5:30, 3], EXP['var/shuffle_gzip_i16'][5:30, 3])


def test_not_hdf5_and_unsupported_files_fail_loudly(tmp_path):
    p = tmp_path / 'x.h5'
    p.write_bytes(b'not an hdf5 file' * 100)
    with pytest.raises(IOError):
        H.File(str(p))
    with pytest.raises(NotImplementedError):
        H.File(os.path.join(HERE, 'vg_sgg_small.h5'), mode='w')
