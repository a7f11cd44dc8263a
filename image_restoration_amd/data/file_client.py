"""Byte sources of the datasets (basicsr/utils/file_client.py): 'disk' reads files, 'lmdb' reads from LMDB environments
(needs the ``lmdb`` module, which this image does not ship: the client then raises at construction)."""


class FileClient:

    def __init__(self, backend='disk', **kwargs):
        self.backend = backend
        if backend == 'disk':
            self._envs = None
        elif backend == 'lmdb':
            try:
                import lmdb
            except ImportError as e:
                raise ImportError('the lmdb io_backend needs the lmdb module') from e
            db_paths, client_keys = kwargs['db_paths'], kwargs.get('client_keys', 'default')
            if isinstance(client_keys, str):
                client_keys = [client_keys]
            if isinstance(db_paths, str):
                db_paths = [db_paths]
            assert len(client_keys) == len(db_paths)
            self._envs = {k: lmdb.open(p, readonly=True, lock=False, readahead=kwargs.get('readahead', False))
                          for k, p in zip(client_keys, db_paths)}
        else:
            raise ValueError(f'Backend {backend} is not supported. Currently supported ones are disk, lmdb')

    def get(self, filepath, client_key='default'):
        if self.backend == 'disk':
            with open(str(filepath), 'rb') as f:
                return f.read()
        with self._envs[client_key].begin(write=False) as txn:
            return txn.get(str(filepath).encode('ascii'))
