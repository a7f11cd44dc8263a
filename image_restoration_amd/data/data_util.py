"""Which LQ file belongs to which GT file (behaviour of basicsr/data/data_util.py:108-231, golden G-s).

Three sources of pairs: two folders (LQ name = ``filename_tmpl.format(<GT stem>) + <GT extension>``), a meta-info file listing GT names,
or two LMDB directories with equal key sets.  All return ``[{'<lq key>_path': ..., '<gt key>_path': ...}, ...]``."""
import os


def scandir(folder):
    """Names of the regular, non-hidden files directly inside ``folder``, sorted (the reference walks in os.scandir order, which
    depends on the file system; sorting yields the same SET of pairs in a reproducible order)."""
    with os.scandir(folder) as it:
        return sorted(entry.name for entry in it if entry.is_file() and entry.name[:1] != '.')


def _two(seq, what):
    if len(seq) != 2:
        raise AssertionError(f'The len of {what} should be 2 with [input_{what[:-1]}, gt_{what[:-1]}]. But got {len(seq)}')
    return seq


def _lq_name(gt_name, filename_tmpl):
    stem, ext = os.path.splitext(os.path.basename(gt_name))
    return filename_tmpl.format(stem) + ext


def _pair(lq_key, gt_key, lq_value, gt_value):
    return {lq_key + '_path': lq_value, gt_key + '_path': gt_value}


def paired_paths_from_folder(folders, keys, filename_tmpl):
    lq_dir, gt_dir = _two(folders, 'folders')
    lq_key, gt_key = _two(keys, 'keys')
    lq_names, gt_names = scandir(lq_dir), scandir(gt_dir)
    if len(lq_names) != len(gt_names):
        raise AssertionError(f'{lq_key} and {gt_key} datasets have different number of images: {len(lq_names)}, {len(gt_names)}.')
    present = frozenset(lq_names)
    pairs = []
    for gt_name in gt_names:
        want = _lq_name(gt_name, filename_tmpl)
        if want not in present:
            raise AssertionError(f'{want} is not in {lq_key}_paths.')
        pairs.append(_pair(lq_key, gt_key, os.path.join(lq_dir, want), os.path.join(gt_dir, gt_name)))
    return pairs


def paired_paths_from_meta_info_file(folders, keys, meta_info_file, filename_tmpl):
    """Each non-empty line of the meta-info file starts with a GT file name (anything after the first blank is ignored)."""
    lq_dir, gt_dir = _two(folders, 'folders')
    lq_key, gt_key = _two(keys, 'keys')
    with open(meta_info_file) as f:
        listed = [ln.split(' ')[0] for ln in (raw.strip() for raw in f) if ln]
    return [_pair(lq_key, gt_key, os.path.join(lq_dir, _lq_name(n, filename_tmpl)), os.path.join(gt_dir, n)) for n in listed]


def lmdb_keys(folder):
    """Keys of an ``*.lmdb`` directory = the part before the first '.' of each ``meta_info.txt`` line (``name.png (h,w,c) level``)."""
    with open(os.path.join(folder, 'meta_info.txt')) as f:
        return [ln.split('.')[0] for ln in f]


def paired_paths_from_lmdb(folders, keys):
    lq_dir, gt_dir = _two(folders, 'folders')
    lq_key, gt_key = _two(keys, 'keys')
    if not (lq_dir.endswith('.lmdb') and gt_dir.endswith('.lmdb')):
        raise ValueError(f'{lq_key} folder and {gt_key} folder should both in lmdb formats. But received '
                         f'{lq_key}: {lq_dir}; {gt_key}: {gt_dir}')
    names = set(lmdb_keys(lq_dir))
    if names != set(lmdb_keys(gt_dir)):
        raise ValueError(f'Keys in {lq_key}_folder and {gt_key}_folder are different.')
    return [_pair(lq_key, gt_key, n, n) for n in sorted(names)]
