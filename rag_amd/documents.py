"""Document — the record the retriever hands back (langchain.schema.Document, imported by the
reference at advanced_rag.py:21 and stringified at advanced_rag.py:123 / :323).
"""
from __future__ import annotations

from typing import Any, Dict, Optional


class Document:
    """page_content + metadata, with the `str()` form LangChain 0.1.x documents have
    (`page_content='…' metadata={…}`), because the reference joins `str(doc)` into the prompt
    context (advanced_rag.py:122-124)."""

    __slots__ = ("page_content", "metadata")

    def __init__(self, page_content: str, metadata: Optional[Dict[str, Any]] = None, **kwargs: Any):
        if not isinstance(page_content, str):
            raise TypeError("page_content must be a str")
        self.page_content = page_content
        self.metadata = dict(metadata) if metadata else {}
        if kwargs:
            self.metadata.update(kwargs.get("extra_metadata", {}))

    def __repr__(self) -> str:
        return f"page_content={self.page_content!r} metadata={self.metadata!r}"

    __str__ = __repr__

    def __eq__(self, other: object) -> bool:
        return (isinstance(other, Document) and self.page_content == other.page_content
                and self.metadata == other.metadata)

    def __hash__(self) -> int:
        return hash(self.page_content)
