import hashlib
import os


def check_integrity(fpath, md5=None):
    if not os.path.isfile(fpath):
        return False
    if md5 is None:
        return True
    with open(fpath, "rb") as fh:
        return hashlib.md5(fh.read()).hexdigest() == md5


def download_and_extract_archive(*args, **kwargs):
    raise RuntimeError("no network in the build container")


def download_url(*args, **kwargs):
    raise RuntimeError("no network in the build container")
