"""Agent-scoped key/value store with buffered writes -- semantics of the reference's
MuJoCo_Gym/data_store.py:1-113 (pinned by Testing/data_store_test.py; its seven cases and a reference-generated trace are checked in tests/test_golden_host.py).

Behaviour that callers can observe and that is kept on purpose:
* writes go to a per-agent buffer and become readable after ``commit()``;
* ``commit()`` leaves the buffer aliased to a shallow copy of the store, so after the first commit a
  write is readable immediately (data_store.py:108);
* ``"global"`` may be selected with ``set_agent`` but cannot be written (data_store.py:44,93-94).

The batched stepper keeps the fused dynamics' per-agent scratch in HBM instead
(``MuJoCoRL.device_store``); this class is the host-side model of it.
"""
from __future__ import annotations


class DataStore:
    def __init__(self, agents):
        self.data = {name: {} for name in agents}
        self.buffer = {name: {} for name in agents}
        self.current_agent = None

    def set_agent(self, agent):
        if agent != "global" and agent not in self.data:
            raise ValueError(f"Agent {agent} is not allowed to write to this dictionary.")
        self.current_agent = agent

    def get_agent_subset(self, agent):
        if agent not in self.data:
            raise ValueError(f"Agent {agent} is not allowed to read from this dictionary.")
        return self.data[agent]

    def _require_agent(self):
        if self.current_agent is None:
            raise ValueError("No agent is currently set.")

    def __getitem__(self, key):
        self._require_agent()
        return self.get_agent_subset(self.current_agent).get(key)

    def __setitem__(self, key, value):
        self._require_agent()
        if self.current_agent not in self.data:
            raise ValueError(f"Agent {self.current_agent} is not allowed to write to this dictionary.")
        self.buffer[self.current_agent][key] = value

    def commit(self):
        for agent, pending in self.buffer.items():
            self.data[agent].update(pending)
        self.buffer = dict(self.data)

    def __repr__(self):
        return repr(self.data)
