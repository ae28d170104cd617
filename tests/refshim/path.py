"""TEST-ONLY stand-in for the third-party ``path`` package (absent in this image),
covering what the reference's tests use: ``tempdir()`` and ``Path``."""
import contextlib
import pathlib
import shutil
import tempfile


class Path(type(pathlib.Path())):
    def rmtree_p(self):
        shutil.rmtree(self, ignore_errors=True)
        return self

    def makedirs_p(self):
        self.mkdir(parents=True, exist_ok=True)
        return self


@contextlib.contextmanager
def tempdir():
    d = tempfile.mkdtemp()
    try:
        yield Path(d)
    finally:
        shutil.rmtree(d, ignore_errors=True)
