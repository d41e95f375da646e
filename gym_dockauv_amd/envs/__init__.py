from .batched import BatchedDocking3d  # noqa: F401
from .docking3d import (BaseDocking3d, CapsuleCurrentDocking3d, CapsuleDocking3d, ObstaclesCurrentDocking3d,  # noqa: F401
                        ObstaclesDocking3d, ObstaclesNoCapDocking3d, SimpleCurrentDocking3d, SimpleDocking3d,
                        SphereDocking3d)
