"""Algorithm-name map (interface of /root/reference/utils/utils.py:5-19), plus `residualtd3`: the residual TD3 that
BASELINE.json names and the reference lacks, composed from the reference's TD3 pieces."""
from ..elegantrl.agent import AgentPPO, AgentTD3
from ..elegantrl.agent_residual import AgentResidualIntegratorModularPPO, AgentResidualPPO, AgentResidualTD3

MODELS = {
    "td3": AgentTD3,
    "ppo": AgentPPO,
    "residualintegratormodularppo": AgentResidualIntegratorModularPPO,
    "residualppo": AgentResidualPPO,
    "residualtd3": AgentResidualTD3,
}
IF_ONPOLICY = {"td3": False, "ppo": True, "residualintegratormodularppo": True, "residualppo": True, "residualtd3": False}
