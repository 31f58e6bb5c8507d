"""Algorithm-name map (interface of /root/reference/utils/utils.py:5-19)."""
from ..elegantrl.agent import AgentPPO, AgentTD3
from ..elegantrl.agent_residual import AgentResidualIntegratorModularPPO, AgentResidualPPO

MODELS = {
    "td3": AgentTD3,
    "ppo": AgentPPO,
    "residualintegratormodularppo": AgentResidualIntegratorModularPPO,
    "residualppo": AgentResidualPPO,
}
IF_ONPOLICY = {"td3": False, "ppo": True, "residualintegratormodularppo": True, "residualppo": True}
