"""Pairing of LQ / GT file paths (basicsr/data/data_util.py:108-231)."""
import os
import os.path as osp


def scandir(folder):
    """Relative paths of the files in ``folder`` (non-recursive, hidden files skipped), sorted."""
    return sorted(e.name for e in os.scandir(folder) if e.is_file() and not e.name.startswith('.'))


def paired_paths_from_folder(folders, keys, filename_tmpl):
    assert len(folders) == 2, f'The len of folders should be 2 with [input_folder, gt_folder]. But got {len(folders)}'
    assert len(keys) == 2, f'The len of keys should be 2 with [input_key, gt_key]. But got {len(keys)}'
    input_folder, gt_folder = folders
    input_key, gt_key = keys
    input_paths, gt_paths = scandir(input_folder), scandir(gt_folder)
    assert len(input_paths) == len(gt_paths), (f'{input_key} and {gt_key} datasets have different number of images: '
                                               f'{len(input_paths)}, {len(gt_paths)}.')
    paths = []
    known = set(input_paths)
    for gt_path in gt_paths:
        basename, ext = osp.splitext(osp.basename(gt_path))
        input_name = f'{filename_tmpl.format(basename)}{ext}'
        assert input_name in known, f'{input_name} is not in {input_key}_paths.'
        paths.append({f'{input_key}_path': osp.join(input_folder, input_name), f'{gt_key}_path': osp.join(gt_folder, gt_path)})
    return paths


def paired_paths_from_meta_info_file(folders, keys, meta_info_file, filename_tmpl):
    assert len(folders) == 2 and len(keys) == 2
    input_folder, gt_folder = folders
    input_key, gt_key = keys
    with open(meta_info_file, 'r') as fin:
        gt_names = [line.strip().split(' ')[0] for line in fin if line.strip()]
    paths = []
    for gt_name in gt_names:
        basename, ext = osp.splitext(osp.basename(gt_name))
        input_name = f'{filename_tmpl.format(basename)}{ext}'
        paths.append({f'{input_key}_path': osp.join(input_folder, input_name), f'{gt_key}_path': osp.join(gt_folder, gt_name)})
    return paths


def paired_paths_from_lmdb(folders, keys):
    """Both folders are ``*.lmdb`` directories with a ``meta_info.txt`` of ``name.png (h,w,c) compress`` lines; the keys of
    the two databases must coincide (data_util.py:108-155)."""
    assert len(folders) == 2 and len(keys) == 2
    input_folder, gt_folder = folders
    input_key, gt_key = keys
    if not (input_folder.endswith('.lmdb') and gt_folder.endswith('.lmdb')):
        raise ValueError(f'{input_key} folder and {gt_key} folder should both in lmdb formats. But received '
                         f'{input_key}: {input_folder}; {gt_key}: {gt_folder}')
    with open(osp.join(input_folder, 'meta_info.txt')) as fin:
        input_keys = [line.split('.')[0] for line in fin]
    with open(osp.join(gt_folder, 'meta_info.txt')) as fin:
        gt_keys = [line.split('.')[0] for line in fin]
    if set(input_keys) != set(gt_keys):
        raise ValueError(f'Keys in {input_key}_folder and {gt_key}_folder are different.')
    return [{f'{input_key}_path': k, f'{gt_key}_path': k} for k in sorted(set(input_keys))]
