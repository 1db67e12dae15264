#include "Buffer.h"

#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>

Buffer::Buffer(size_t len, const void *data) { setData(data, len); }

Buffer::Buffer(const Buffer &other) { setData(other.m_data, other.m_len); }

Buffer::Buffer(Buffer &&other) noexcept
    : m_len(other.m_len), m_capacity(other.m_capacity), m_data(other.m_data)
{
    other.m_len = other.m_capacity = 0;
    other.m_data = nullptr;
}

Buffer::Buffer(const std::vector<uint8_t> &vec) { setData(vec.data(), vec.size()); }

Buffer::~Buffer() { std::free(m_data); }

void Buffer::swap(Buffer &o) noexcept
{
    std::swap(m_len, o.m_len);
    std::swap(m_capacity, o.m_capacity);
    std::swap(m_data, o.m_data);
}

void swap(Buffer &a, Buffer &b) noexcept { a.swap(b); }

void Buffer::setLength(size_t len)
{
    if (len > m_capacity) {
        void *fresh = nullptr;
        const int rc = posix_memalign(&fresh, 32, len);
        if (rc != 0) throw std::runtime_error("memory allocation failed: " + std::to_string(rc));
        if (m_data) {
            std::memcpy(fresh, m_data, m_len);
            std::free(m_data);
        }
        m_data = fresh;
        m_capacity = len;
    }
    m_len = len;
}

void Buffer::setData(const void *data, size_t len)
{
    setLength(0);
    appendData(data, len);
}

void Buffer::appendData(const void *data, size_t len)
{
    const size_t at = m_len;
    setLength(at + len);
    if (data && len) std::memcpy(static_cast<uint8_t *>(m_data) + at, data, len);
}

Buffer &Buffer::operator=(const Buffer &other)
{
    if (this != &other) setData(other.m_data, other.m_len);
    return *this;
}

Buffer &Buffer::operator=(Buffer &&other) noexcept
{
    if (this != &other) {
        std::free(m_data);
        m_len = other.m_len;
        m_capacity = other.m_capacity;
        m_data = other.m_data;
        other.m_len = other.m_capacity = 0;
        other.m_data = nullptr;
    }
    return *this;
}

Buffer &Buffer::operator=(const std::vector<uint8_t> &buf)
{
    setData(buf.data(), buf.size());
    return *this;
}

Buffer &Buffer::operator+=(const Buffer &other)
{
    appendData(other.m_data, other.m_len);
    return *this;
}

uint8_t Buffer::operator[](size_t i) const
{
    if (i >= m_len) throw std::out_of_range("index out of range");
    return static_cast<const uint8_t *>(m_data)[i];
}
