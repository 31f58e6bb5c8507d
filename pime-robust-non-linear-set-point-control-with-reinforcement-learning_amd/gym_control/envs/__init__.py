from .ph import (PH1DChangingParamUniformGoalIntegrator,  # noqa: F401
                 PH1DChangingParamUniformGoalIntegrator_NoBound)
from .nonlinear_watertank import (NonLinearWaterTankChangingParamUniformGoal,  # noqa: F401
                                  NonLinearWaterTankChangingParamUniformGoalIntegrator,
                                  NonLinearWaterTankChangingParamUniformGoalStacking)
