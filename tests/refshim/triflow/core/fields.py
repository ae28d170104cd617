from triflow_amd.fields import BaseFields                   # noqa: F401
