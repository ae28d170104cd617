"""triflow_amd: MI355X-native F/J evaluation and implicit time stepping for
1-D finite-difference PDE models, behind the reference's (celliern/triflow)
``Model`` / ``Simulation`` / ``schemes`` interface."""

import logging

from .model import Model                       # noqa: F401
from .fields import BaseFields                 # noqa: F401
from . import schemes                          # noqa: F401
from .simulation import Simulation             # noqa: F401
from .container import TriflowContainer as Container, retrieve_container   # noqa: F401

logging.getLogger(__name__).addHandler(logging.NullHandler())

__all__ = ["Model", "Simulation", "schemes", "BaseFields", "Container", "retrieve_container"]
