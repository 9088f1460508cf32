"""Host-side mirror of the reference's ``pcdet`` package for the Det6D inference path
(core/pcdet/__init__.py).  Put ``de6d_amd`` on sys.path to ``import pcdet`` exactly like the
reference, or import ``de6d_amd.pcdet`` directly."""
__version__ = "0.5.2+det6d.mi355x"
