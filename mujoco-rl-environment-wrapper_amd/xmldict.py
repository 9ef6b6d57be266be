"""XML text -> nested dict in the layout the reference's table builders walk.

The reference loads its level with ``xmltodict.parse`` (mujoco_parent.py:92-94) and every
index table (gather / scatter) is built by walking that dict in document order
(mujoco_parent.py:618-648).  ``xmltodict`` is not installed in this image, so this module
produces the same structure from the standard library parser:

* attributes become ``"@name"`` keys, inserted before child keys,
* a tag that occurs once under its parent maps to a dict (or ``None`` when it has neither
  attributes nor children), a tag that occurs several times maps to a list, kept at the
  position of the first occurrence,
* the document root is wrapped as ``{root_tag: ...}``.
"""
from __future__ import annotations

import xml.etree.ElementTree as ET


def _convert(elem: ET.Element):
    out = {}
    for key, value in elem.attrib.items():
        out["@" + key] = value
    for child in elem:
        item = _convert(child)
        tag = child.tag
        if tag in out:
            if isinstance(out[tag], list):
                out[tag].append(item)
            else:
                out[tag] = [out[tag], item]
        else:
            out[tag] = item
    text = (elem.text or "").strip()
    if text:
        if out:
            out["#text"] = text
        else:
            return text
    if not out:
        return None
    return out


def parse(text: str) -> dict:
    root = ET.fromstring(text)
    return {root.tag: _convert(root)}


def find_in_nested_dict(dictionary, name=None, filter_key="@name", parent=None) -> list:
    """Depth-first search with the semantics of mujoco_parent.py:618-648.

    ``parent=tag`` collects the children stored under ``tag`` (a list contributes each item, a
    single dict contributes itself), optionally filtered by ``item[filter_key] == name``.
    Without ``parent`` it collects every dict holding ``filter_key == name``.  Results come out
    in document order; nested matches are included (a matching dict is still descended into).
    """
    found = []
    if isinstance(dictionary, dict):
        if parent is not None and parent in dictionary:
            node = dictionary[parent]
            if isinstance(node, list):
                for item in node:
                    if not name or (filter_key in item and item[filter_key] == name):
                        found.append(item)
            elif not name or node[filter_key] == name:
                found.append(node)
        for key, value in dictionary.items():
            if (key == filter_key or not filter_key) and (value == name or not name) and not parent:
                found.append(dictionary)
            elif isinstance(value, (dict, list)):
                found.extend(find_in_nested_dict(value, name, filter_key=filter_key, parent=parent))
    elif isinstance(dictionary, list):
        for item in dictionary:
            if isinstance(item, (dict, list)):
                found.extend(find_in_nested_dict(item, name, filter_key=filter_key, parent=parent))
    return found
